"""CTS with the mixture-of-experts student encoder (rsl_rl/rsl_rl/algorithms/moe_cts.py:40-234): the student step adds a
load-balance term  coef * mean((mean_batch(gate) - 1/E)^2)  (:205-209); update() also returns its mean."""
import torch

from .cts import CTS


class MoECTS(CTS):
    _NUM_STUDENT_LOGS = 2

    def __init__(self, model, num_envs, history_length, load_balance_coef=0.01, **kwargs):
        super().__init__(model, num_envs, history_length, **kwargs)
        self.load_balance_coef = load_balance_coef

    def _student_losses(self, hist_s, priv_s):
        student_latent, gate = self.model.student_latent(hist_s)
        with torch.no_grad():
            teacher_latent = self.model.teacher_encoder(priv_s)
        latent_loss = (teacher_latent - student_latent).pow(2).mean()
        usage = gate.mean(dim=0)
        load_balance_loss = (usage - 1.0 / gate.shape[1]).pow(2).mean()
        return latent_loss + self.load_balance_coef * load_balance_loss, (latent_loss, load_balance_loss)


class MoENGCTS(MoECTS):
    """rsl_rl/rsl_rl/algorithms/moe_ng_cts.py: MoECTS whose student latent comes from the no-goal encoder (the model's `student_latent` hook)."""
