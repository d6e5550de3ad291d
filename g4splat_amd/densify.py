"""Adaptive density control of the Gaussian set: the build's counterpart of
2dgs/scene/gaussian_model.py:436-651 -- reset_opacity, prune_points, densify_and_clone, densify_and_split,
densify_and_prune, compute_mip_filter (:388-434) -- with the same selection rules, the same new-point
construction and the same optimiser-state surgery (moments of kept rows survive, new rows start at zero), written
once around a single row-edit primitive instead of three copies of the group loop.  On the HIP device the row
filtering is stream compaction in the library (optim.compact_rows -> g4s_compact_scan / g4s_compact_gather: wave
ballot + prefix sum, the mask scanned once for all eighteen tensors); host tensors take torch indexing, which is
what tests/test_densify.py replays against the reference's own GaussianModel.  Works with torch.optim.Adam and
optim.FusedAdam alike because both keep `state[p] = {step, exp_avg, exp_avg_sq}`.

Mixed into GaussianModel (gaussian_model.py imports DensifyMixin)."""
import torch
from torch import nn

_FIELDS = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
           "scaling": "_scaling", "rotation": "_rotation"}


def build_rotation(q):
    """2dgs/utils/general_utils.py:79-101: rotation matrices of (w, x, y, z) quaternions, normalised first."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


class DensifyMixin:
    percent_dense = 0.01  # arguments/__init__.py: OptimizationParams.percent_dense

    # ---- the one primitive: keep a subset of rows and / or append rows, in the parameters AND in the optimiser ----
    def _edit_rows(self, keep=None, append=None, zero_moments=False, also=()):
        """keep: bool[P] or None; append: dict group-name -> new rows or None; also: further [P, ...] tensors to filter
        with `keep` (the densification statistics).  Rebinds the six parameters (gaussian_model.py:495-560:
        replace_tensor_to_optimizer / _prune_optimizer / cat_tensors_to_optimizer) and returns the filtered `also`.

        HIP tensors: `keep` is scanned once and all eighteen tensors (+ `also`) are compacted against that scan
        (optim.compact_rows: wave ballot + prefix sum, three gather launches); host tensors take torch indexing."""
        groups = [g for g in self.optimizer.param_groups if g["name"] in _FIELDS]
        for g in groups:
            assert len(g["params"]) == 1
        olds = [g["params"][0] for g in groups]
        states = [self.optimizer.state.pop(o, None) for o in olds]
        has_state = [st is not None and len(st) > 0 for st in states]
        extras = [append[g["name"]].detach() if append is not None else None for g in groups]
        # every tensor that has to be filtered, in one list
        work = [o.detach() for o in olds]
        for st, ok in zip(states, has_state):
            if ok:
                work += [st["exp_avg"], st["exp_avg_sq"]]
        n_also = len(also)
        work += [a.float() if a.dtype != torch.float32 else a for a in also]
        if keep is None:
            filtered = work
        elif work[0].is_cuda:
            from .optim import compact_rows
            _n, filtered = compact_rows(keep, work)
        else:
            keep_idx = torch.nonzero(keep).squeeze(1)  # one mask -> index conversion for all tensors
            filtered = [t.index_select(0, keep_idx) for t in work]
        it = iter(filtered)
        datas = [next(it) for _ in olds]
        moments = [(next(it), next(it)) if ok else None for ok in has_state]
        also_out = [next(it) for _ in range(n_also)]
        for g, old, st, ok, data, mom, extra in zip(groups, olds, states, has_state, datas, moments, extras):
            if extra is not None:
                data = torch.cat((data, extra), dim=0)
            new = nn.Parameter(data.contiguous().requires_grad_(True))
            if ok:
                for key, m in zip(("exp_avg", "exp_avg_sq"), mom):
                    if extra is not None:
                        m = torch.cat((m, torch.zeros_like(extra)), dim=0)
                    st[key] = torch.zeros_like(new) if zero_moments else m.contiguous()
                self.optimizer.state[new] = st
            g["params"][0] = new
            setattr(self, _FIELDS[g["name"]], new)
        return also_out

    @staticmethod
    def _pick(sel, tensors):
        """[t[sel] for t in tensors]: one mask scan + one gather launch on the HIP device, torch indexing on the host."""
        ts = [t.detach() for t in tensors]
        if ts[0].is_cuda:
            from .optim import compact_rows
            return compact_rows(sel, ts)[1]
        i = torch.nonzero(sel).squeeze(1)
        return [t.index_select(0, i) for t in ts]

    # ---- reference-named entry points --------------------------------------------------------------------------
    def replace_tensor_to_optimizer(self, tensor, name):
        """:495-508: swap one group's tensor, zeroing its moments."""
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                old = group["params"][0]
                state = self.optimizer.state.pop(old, None)
                new = nn.Parameter(tensor.detach().clone().requires_grad_(True))
                if state is not None and len(state) > 0:
                    state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
                    self.optimizer.state[new] = state
                group["params"][0] = new
                setattr(self, _FIELDS[name], new)
                return {name: new}
        raise KeyError(name)

    def _prune_optimizer(self, mask):
        """:510-525 (mask = rows to KEEP); returns the rebound parameters by group name."""
        self._edit_rows(keep=mask)
        return {g["name"]: g["params"][0] for g in self.optimizer.param_groups}

    def cat_tensors_to_optimizer(self, tensors_dict):
        """:543-560; returns the rebound parameters by group name."""
        self._edit_rows(append=tensors_dict)
        return {g["name"]: g["params"][0] for g in self.optimizer.param_groups}

    def reset_opacity(self):
        """:436-439: opacities above 0.01 are pulled down to 0.01 (through the activation's inverse)."""
        cur = self.get_opacity
        new = torch.log((m := torch.minimum(cur, torch.full_like(cur, 0.01))) / (1 - m))
        self.replace_tensor_to_optimizer(new, "opacity")

    def prune_points(self, mask):
        """:527-541: drop the rows where `mask` is True, everywhere (parameters, moments, statistics)."""
        keep = ~mask
        stats = [self.xyz_gradient_accum, self.denom, self.max_radii2D]
        has_mip = self.mip_filter is not None and self.mip_filter.shape[0] == keep.shape[0]
        if has_mip:
            stats.append(self.mip_filter)
        out = self._edit_rows(keep=keep, also=stats)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = out[0], out[1], out[2]
        if has_mip:
            self.mip_filter = out[3]

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation):
        """:562-581: append rows; the densification statistics restart from zero for everybody."""
        self._edit_rows(append={"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest,
                                "opacity": new_opacities, "scaling": new_scaling, "rotation": new_rotation})
        n, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """:612-626: small Gaussians with a large screen-space gradient are duplicated in place."""
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(*self._pick(sel, [self._xyz, self._features_dc, self._features_rest, self._opacity,
                                                     self._scaling, self._rotation]))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        """:583-610: large Gaussians with a large gradient are replaced by N children sampled inside them (in the
        splat's own plane: the third standard deviation is 0), each 1/(0.8 N) the size.  `generator` (extension): the
        torch.Generator the children are drawn from (:598 draws from the process-wide one, which is what None does) --
        under data parallelism every replica passes one seeded from the iteration, parallel.densify_generator, so that
        all of them draw the same children."""
        P = self._xyz.shape[0]
        padded = torch.zeros((P,), device=self._xyz.device)
        padded[:grads.shape[0]] = grads.squeeze(-1) if grads.ndim > 1 else grads
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        p_scal, p_rot, p_xyz, p_dc, p_rest, p_opa = self._pick(sel, [self.get_scaling, self._rotation, self._xyz,
                                                                     self._features_dc, self._features_rest, self._opacity])
        s = p_scal.repeat(N, 1)
        stds = torch.cat([s, torch.zeros_like(s[:, :1])], dim=-1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        R = build_rotation(p_rot).repeat(N, 1, 1)
        new_xyz = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + p_xyz.repeat(N, 1)
        new_scaling = torch.log(s / (0.8 * N))
        self.densification_postfix(new_xyz, p_dc.repeat(N, 1, 1), p_rest.repeat(N, 1, 1), p_opa.repeat(N, 1), new_scaling,
                                   p_rot.repeat(N, 1))
        gone = torch.cat((sel, torch.zeros(N * int(p_xyz.shape[0]), dtype=torch.bool, device=sel.device)))
        self.prune_points(gone)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """:628-647.  The mip filter is switched off while sizes and opacities are compared, like the reference.
        `generator`: see densify_and_split."""
        mip = self.use_mip_filter
        self.use_mip_filter = False
        try:
            grads = self.xyz_gradient_accum / self.denom
            grads[grads.isnan()] = 0.0
            self.densify_and_clone(grads, max_grad, extent)
            self.densify_and_split(grads, max_grad, extent, generator=generator)
            prune = (self.get_opacity < min_opacity).squeeze(-1)
            if max_screen_size:
                prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
            self.prune_points(prune)
        finally:
            self.use_mip_filter = mip

    @torch.no_grad()
    def compute_mip_filter(self, cameras, znear=0.2, filter_variance=0.2):
        """:388-434: per-Gaussian low-pass size = (depth of the closest camera that sees it) / (largest focal length)
        * sqrt(filter_variance).  Cameras expose R, T, focal_x, focal_y, image_width, image_height."""
        xyz = self._xyz
        distance = torch.full((xyz.shape[0],), 100000.0, device=xyz.device)
        seen = torch.zeros((xyz.shape[0],), dtype=torch.bool, device=xyz.device)
        focal = 0.0
        for cam in cameras:
            R = torch.as_tensor(cam.R, device=xyz.device, dtype=torch.float32)
            T = torch.as_tensor(cam.T, device=xyz.device, dtype=torch.float32)
            pc = xyz @ R + T[None, :]
            z = pc[:, 2].clamp(min=0.001)
            u = pc[:, 0] / z * cam.focal_x + cam.image_width / 2.0
            v = pc[:, 1] / z * cam.focal_y + cam.image_height / 2.0
            ok = (pc[:, 2] > znear) & (u >= -0.15 * cam.image_width) & (u <= 1.15 * cam.image_width) & \
                 (v >= -0.15 * cam.image_height) & (v <= 1.15 * cam.image_height)
            distance[ok] = torch.minimum(distance[ok], z[ok])
            seen |= ok
            focal = max(focal, cam.focal_x)
        distance[~seen] = distance[seen].max()
        self.mip_filter = (distance / focal * (filter_variance ** 0.5))[..., None]
