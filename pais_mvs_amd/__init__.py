"""pais_mvs_amd -- MI355X-native (gfx950) patch refinement + expansion hot path of PAIS-MVS.

Host-side mirror of the reference's interface for this path (MvsConfig, Camera,
Patch candidates, MVS::refineSeedPatches / MVS::expansionPatches) over the C ABI
in include/pais_hip.h.  See DESIGN.md.
"""
from .config import MvsConfig, readme_config, default_config  # noqa: F401
from .camera import Camera  # noqa: F401
from .context import Context  # noqa: F401
