from .helpers import class_to_dict, get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args, SimParams  # noqa: F401


def __getattr__(name):   # task_registry imports the runners, which import utils.helpers: resolve lazily
    if name == "task_registry":
        from .task_registry import task_registry
        return task_registry
    raise AttributeError(name)
