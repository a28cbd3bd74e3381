from .remap import overlap_remap, overlap_remap_packed, RemapResult, revoke_map  # noqa: F401
from .interaction import Interaction  # noqa: F401
from .dataloader import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader, FullSortEvalLoader  # noqa: F401
