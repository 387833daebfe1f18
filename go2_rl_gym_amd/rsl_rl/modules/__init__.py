from .actor_critic import ActorCritic, get_activation
