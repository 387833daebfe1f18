from .actor_critic import ActorCritic, get_activation
from .actor_critic_cts import ActorCriticCTS
from .actor_critic_moe_cts import ActorCriticMoECTS
from .actor_critic_moe_ng_cts import ActorCriticMoENGCTS
from .actor_critic_ac_moe_cts import ActorCriticACMoECTS, ActorCriticDualMoECTS
