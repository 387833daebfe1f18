from .rollout_storage import RolloutStorage
from .rollout_storage_cts import RolloutStorageCTS
