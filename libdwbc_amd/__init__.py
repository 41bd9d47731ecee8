"""libdwbc_amd -- MI355X-native batched HQP whole-body-control hot path behind libdwbc's RobotData API.

The product is libdwbc_hip.so (hand-written HIP for gfx950 + a C-ABI, include/dwbc_batch.h).  This package is the
thin Python host layer over that ABI.  It fails loudly when the library or a GPU is missing.
"""
from .batch import (  # noqa: F401
    CONTACT_6D, TASK_LINK_6D, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME, TASK_LINK_POSITION,
    TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME, TASK_LINK_ROTATION, TASK_LINK_ROTATION_CUSTOM_FRAME,
    Batch, DwbcError, Model, build_pack,
)

from .rl_bridge import RlWBCBridge  # noqa: F401,E402
from .hqp import HQP  # noqa: F401,E402

__all__ = ["Batch", "Model", "DwbcError", "RlWBCBridge", "HQP", "build_pack"]
