"""CTS with the mixture-of-experts student encoder (rsl_rl/rsl_rl/algorithms/moe_cts.py:40-234): the student step adds a
load-balance term  coef * mean((mean_batch(gate) - 1/E)^2)  (:205-209); update() also returns its mean."""
import torch

from .cts import CTS


class MoECTS(CTS):
    _NUM_STUDENT_LOGS = 2

    def __init__(self, model, num_envs, history_length, load_balance_coef=0.01, **kwargs):
        super().__init__(model, num_envs, history_length, **kwargs)
        self.load_balance_coef = load_balance_coef

    def _student_losses(self, hist_s, priv_s, teacher_latent=None):
        student_latent, gate = self.model.student_latent(hist_s)
        teacher_latent = self._teacher_latent(priv_s, teacher_latent)
        latent_loss = (teacher_latent - student_latent).pow(2).mean()
        usage = gate.mean(dim=0)
        load_balance_loss = (usage - 1.0 / gate.shape[1]).pow(2).mean()
        return latent_loss + self.load_balance_coef * load_balance_loss, (latent_loss, load_balance_loss)


    def _student_backward(self, hist_s, priv_s, teacher_latent):
        """Graph mode on the library pair: the mixture, the normaliser, both losses and their backward as two launches (modules/fused_cts.py:moe_head_grads) — autograd
        only runs through the gate MLP, the experts' backbone and the expert heads, seeded with the head's analytic gradients."""
        from ..modules import fused_cts
        t_hat = self._teacher_latent(priv_s, teacher_latent)
        if not (self.fused_loss and fused_cts.moe_head_applicable(self.model, t_hat.shape[1])):
            return super()._student_backward(hist_s, priv_s, teacher_latent)
        logits, outs, bias = self.model.student_moe_parts(hist_s)
        _, dl, do, dbias = fused_cts.moe_head_grads(logits, outs, t_hat, self.load_balance_coef, acc=self._acc[3 + self._NUM_POLICY_LOGS:], expert_major=True, bias=bias)
        torch.autograd.backward([logits, outs], [dl, do])
        bias.grad = dbias.view_as(bias)          # (the heads' bias is added inside the loss head: its gradient comes from there, not through autograd)


class MoENGCTS(MoECTS):
    """rsl_rl/rsl_rl/algorithms/moe_ng_cts.py: MoECTS whose student latent comes from the no-goal encoder (the model's `student_latent` hook)."""


def _balance(weights):
    """mean((mean_batch(gate) - 1/E)^2) (ac_moe_cts.py:186-189)"""
    return (weights.mean(dim=0) - 1.0 / weights.shape[1]).pow(2).mean()


class ACMoECTS(CTS):
    """CTS with the mixture-of-experts actor / expert critic (rsl_rl/rsl_rl/algorithms/ac_moe_cts.py:40-250): the policy loss adds
    coef * load-balance of the ACTOR gate over the whole [teacher | student] mini-batch; update() returns it after the latent loss.
    compute_returns takes the proprioceptive observation too (the value head mixes its experts with the actor gate)."""
    _NUM_POLICY_LOGS = 1

    def __init__(self, model, num_envs, history_length, load_balance_coef=0.01, **kwargs):
        super().__init__(model, num_envs, history_length, **kwargs)
        self.load_balance_coef = load_balance_coef

    def _policy_extra(self, loss, aux):
        lb = _balance(aux)
        self._policy_logs = (lb.detach(),)
        return loss + self.load_balance_coef * lb

    def compute_returns(self, last_obs, last_privileged_obs, last_history):
        super().compute_returns(last_privileged_obs, last_history, last_obs)


class DualMoECTS(ACMoECTS):
    """AC-MoE heads + the MoE student encoder (dual_moe_cts.py:40-262): returns (..., latent, student load balance, actor load balance)."""
    _NUM_STUDENT_LOGS = 2

    def _student_losses(self, hist_s, priv_s, teacher_latent=None):
        student_latent, gate = self.model.student_latent(hist_s)
        teacher_latent = self._teacher_latent(priv_s, teacher_latent)
        latent_loss = (teacher_latent - student_latent).pow(2).mean()
        lb = _balance(gate)
        return latent_loss + self.load_balance_coef * lb, (latent_loss, lb)


class MCPCTS(CTS):
    """rsl_rl/rsl_rl/algorithms/mcp_cts.py: CTS on the multiplicative-compositional actor (3 param groups: teacher encoder, critic, actor_mcp)."""
