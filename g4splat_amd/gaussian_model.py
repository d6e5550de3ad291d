"""Minimal GaussianModel-compatible parameter container: exactly the attributes `render()` and the
training loop read (2dgs/scene/gaussian_model.py:157-266) -- getters with the reference's activations,
`create_from_pcd` (distCUDA2 scale init), `create_from_parameters`, Adam groups with the reference's
names / eps, the densification statistics, and the reference's PLY format (save_ply / load_ply via
ply_io.py); adaptive density control (prune / clone / split / reset_opacity / mip filter) comes from
densify.py."""
import torch
from torch import nn

from .densify import DensifyMixin

C0 = 0.28209479177387814


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), optionally eased in by a sine
    ramp from lr_delay_mult over the first lr_delay_steps (2dgs/utils/general_utils.py:30-64; pinned by
    tests/golden/lr_schedule.npz)."""
    import math

    def lr(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        ramp = 1.0
        if lr_delay_steps > 0:
            ramp = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        return ramp * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    return lr


class GaussianModel(DensifyMixin):
    def __init__(self, sh_degree=3, use_mip_filter=False):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.use_mip_filter = use_mip_filter
        self.mip_filter = None
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer = None
        self.spatial_lr_scale = 1.0

    # ---- activations / getters (gaussian_model.py:157-192) ----------------------------------
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        s = torch.exp(self._scaling)
        if self.use_mip_filter:
            s = torch.sqrt(torch.square(s) + torch.square(self.mip_filter))
        return s

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_features_split(self):
        """(features_dc [P,1,3], features_rest [P,M-1,3]) for the rasterizer's split-SH entry points: what
        `get_features` concatenates, handed over in place (extension over the reference)."""
        return (self._features_dc, self._features_rest)

    @property
    def get_opacity(self):
        o = torch.sigmoid(self._opacity)
        if self.use_mip_filter:
            s2 = torch.square(torch.exp(self._scaling))
            o = o * torch.sqrt(s2.prod(dim=1) / (s2 + torch.square(self.mip_filter)).prod(dim=1))[..., None]
        return o

    @property
    def get_activated(self):
        """(get_scaling, get_rotation, get_opacity) at once: one fused HIP launch each way on a HIP device with the mip
        filter off (optim.fused_activations), the three getters otherwise.  render() reads this when present."""
        if self._scaling.is_cuda and not self.use_mip_filter:
            from .optim import fused_activations
            return fused_activations(self._scaling, self._rotation, self._opacity)
        return self.get_scaling, self.get_rotation, self.get_opacity

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    # ---- construction ------------------------------------------------------------------------------
    def _set(self, xyz, colors, log_scales, rots, opacity=0.1):
        n, dev = xyz.shape[0], xyz.device
        feats = torch.zeros((n, 3, (self.max_sh_degree + 1) ** 2), device=dev)
        feats[:, :3, 0] = (colors - 0.5) / C0
        self._xyz = nn.Parameter(xyz.clone().float().requires_grad_(True))
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(log_scales.float().requires_grad_(True))
        self._rotation = nn.Parameter(rots.float().requires_grad_(True))
        self._opacity = nn.Parameter(inverse_sigmoid(opacity * torch.ones((n, 1), device=dev)).requires_grad_(True))
        self.max_radii2D = torch.zeros(n, device=dev)

    def create_from_pcd(self, points, colors, spatial_lr_scale=1.0):
        """gaussian_model.py:201-223: isotropic scale = sqrt(mean squared distance to the 3 nearest points)."""
        from .simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        dist2 = torch.clamp_min(distCUDA2(points.float()), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
        rots = torch.rand((points.shape[0], 4), device=points.device)
        self._set(points, colors, scales, rots)

    def create_from_parameters(self, means, scales, quaternions, colors, spatial_lr_scale=1.0):
        """gaussian_model.py:225-246 (the path G4Splat's trainer uses)."""
        self.spatial_lr_scale = spatial_lr_scale
        self._set(means, colors, torch.log(scales), quaternions)

    # ---- optimisation ------------------------------------------------------------------------------
    def training_setup(self, training_args=None, position_lr=0.00016, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                       rotation_lr=0.001, fused=None, capturable=False):
        """gaussian_model.py:248-266.  `training_args`: the reference's OptimizationParams-like object
        (position_lr_init/final/delay_mult/max_steps, feature_lr, opacity_lr, scaling_lr, rotation_lr, percent_dense);
        without it the keyword defaults (= arguments/__init__.py) apply.  `fused` (default: on for HIP tensors):
        one-kernel Adam (optim.FusedAdam) instead of torch.optim.Adam's foreach passes; same groups, names, lr, eps and
        state layout.  `capturable`: step counts and learning rates on the device (for graphed.TrainStepGraph)."""
        position_lr_final, delay_mult, max_steps = position_lr / 100.0, 0.01, 30000
        if training_args is not None:
            ta = training_args
            position_lr, feature_lr, opacity_lr = ta.position_lr_init, ta.feature_lr, ta.opacity_lr
            scaling_lr, rotation_lr = ta.scaling_lr, ta.rotation_lr
            position_lr_final, delay_mult, max_steps = ta.position_lr_final, ta.position_lr_delay_mult, ta.position_lr_max_steps
            self.percent_dense = ta.percent_dense
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=position_lr * self.spatial_lr_scale,
                                                    lr_final=position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=delay_mult, max_steps=max_steps)
        n, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        groups = [
            {"params": [self._xyz], "lr": position_lr * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": rotation_lr, "name": "rotation"},
        ]
        if fused is None:
            fused = self._xyz.is_cuda
        if fused:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15, capturable=capturable)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, capturable=capturable)

    def update_learning_rate(self, iteration):
        """gaussian_model.py:268-274: the xyz group follows the exponential schedule; returns the new rate."""
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
                return group["lr"]

    def capture(self):
        """gaussian_model.py:65-79: the checkpoint tuple train_with_refine_depth.py:606-608 hands to torch.save."""
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale)

    def restore(self, model_args, training_args=None, **setup_kw):
        """gaussian_model.py:81-97."""
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity, self.max_radii2D, accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        self.training_setup(training_args, **setup_kw)
        self.xyz_gradient_accum, self.denom = accum, denom
        self.optimizer.load_state_dict(opt_dict)

    def freeze_params(self):
        """gaussian_model.py:99-121: no parameter receives gradients any more."""
        for p in self.parameters():
            p.requires_grad_(False)

    def get_covariance(self, scaling_modifier=1):
        """gaussian_model.py:29-36,193-194: the 4x4 splat-to-world transform per Gaussian (rows: the two scaled tangent
        axes, the normal, the centre) that render()'s compute_cov3D_python path multiplies into world2pix."""
        from .densify import build_rotation
        s = self.get_scaling * scaling_modifier
        R = build_rotation(self._rotation)
        L = R * torch.cat([s, torch.ones_like(s)], dim=-1)[:, None, :3]  # R @ diag(sx, sy, 1)
        trans = torch.zeros((self._xyz.shape[0], 4, 4), dtype=torch.float, device=self._xyz.device)
        trans[:, :3, :3] = L.permute(0, 2, 1)
        trans[:, 3, :3] = self.get_xyz
        trans[:, 3, 3] = 1
        return trans

    def set_mip_filter(self, use_mip_filter):
        self.use_mip_filter = bool(use_mip_filter)

    def gs_scale_loss(self, max_scale_thresh=0.05):
        """gaussian_model.py:653-657: squared excess of the larger axis over a threshold, summed."""
        excess = torch.clamp(self.get_scaling.max(dim=1).values - max_scale_thresh, min=0.0)
        return torch.sum(excess ** 2)

    def construct_list_of_attributes(self):
        """gaussian_model.py:276-291."""
        from .ply_io import attribute_names
        return attribute_names(self._features_rest.shape[1], self._scaling.shape[1], self._rotation.shape[1],
                               self.use_mip_filter)

    def add_densification_stats(self, viewspace_point_tensor, update_filter, radii=None):
        """gaussian_model.py:649-651: accumulate the per-view norm of the screen-space gradient.  `radii` (extension):
        also fold the training loop's `max_radii2D[filter] = max(max_radii2D[filter], radii[filter])` in.  On a HIP
        device this is one fused kernel (optim.densify_stats); host tensors take the reference's torch formulation."""
        grad = viewspace_point_tensor.grad
        if grad.is_cuda:
            from .optim import densify_stats
            densify_stats(grad, update_filter, self.xyz_gradient_accum, self.denom, radii,
                          self.max_radii2D if radii is not None else None)
            return
        self.xyz_gradient_accum[update_filter] += torch.norm(grad[update_filter], dim=-1, keepdim=True)
        self.denom[update_filter] += 1
        if radii is not None:
            self.max_radii2D[update_filter] = torch.max(self.max_radii2D[update_filter], radii[update_filter].float())

    # ---- on-disk format (gaussian_model.py:293-315, 441-493) ----------------------------------
    def save_ply(self, path):
        from .ply_io import write_gaussian_ply
        c = lambda t: t.detach().cpu().numpy()
        write_gaussian_ply(path, c(self._xyz), c(self._features_dc), c(self._features_rest), c(self._opacity),
                           c(self._scaling), c(self._rotation),
                           c(self.mip_filter) if self.use_mip_filter and self.mip_filter is not None else None)

    def load_ply(self, path, device=None):
        """Loads a scene written by the reference (or by save_ply).  `device` defaults to "cuda" like the
        reference (:484-491)."""
        from .ply_io import read_gaussian_ply
        d = read_gaussian_ply(path, self.max_sh_degree)
        dev = torch.device(device if device is not None else "cuda")
        par = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=dev).requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = par(d["xyz"]), par(d["features_dc"]), par(d["features_rest"])
        self._opacity, self._scaling, self._rotation = par(d["opacity"]), par(d["scaling"]), par(d["rotation"])
        self.use_mip_filter = d["mip_filter"] is not None
        self.mip_filter = torch.tensor(d["mip_filter"], dtype=torch.float, device=dev) if self.use_mip_filter else None
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=dev)
        self.active_sh_degree = self.max_sh_degree
