from .on_policy_runner import OnPolicyRunner
from .on_policy_runner_cts import OnPolicyRunnerCTS
