"""Go2Robot (legged_gym/envs/go2/go2_env.py): the 45-d observation / 263-d privileged observation layout, its noise
vector (:9-21) and the two hip rewards (:55-68) are implemented inside the kernels (csrc/go2_post.h); this subclass
only documents the layout and exposes the noise vector under the reference's attribute name."""
import torch

from ..base.legged_robot import LeggedRobot


class Go2Robot(LeggedRobot):
    def _init_buffers(self):
        super()._init_buffers()
        self.noise_scale_vec = self._get_noise_scale_vec(self.cfg)

    def _get_noise_scale_vec(self, cfg):
        n, o = cfg.noise, self.obs_scales
        v = torch.zeros(self.num_obs, device=self.device)
        v[0:3] = n.noise_scales.ang_vel * n.noise_level * o.ang_vel
        v[3:6] = n.noise_scales.gravity * n.noise_level
        v[9:21] = n.noise_scales.dof_pos * n.noise_level * o.dof_pos
        v[21:33] = n.noise_scales.dof_vel * n.noise_level * o.dof_vel
        return v
