"""Golden vectors from the reference's importable Python helpers that are not covered by make_golden.py:

    lr_schedule.npz   2d-gaussian-splatting/utils/general_utils.py:get_expon_lr_func  (the xyz learning-rate
                      schedule of GaussianModel.update_learning_rate, scene/gaussian_model.py:262-274)

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_misc.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/2d-gaussian-splatting"


def main():
    sys.path.insert(0, REF)
    from utils import general_utils as gu
    steps = np.array([-1, 0, 1, 10, 100, 999, 1000, 7000, 15000, 29999, 30000, 45000], np.int64)
    cases = [(0.00016 * 4.3, 0.0000016 * 4.3, 0, 0.01, 30000), (1e-2, 1e-4, 500, 0.1, 20000), (0.0, 0.0, 0, 1.0, 1000)]
    out = {"steps": steps}
    for i, (lr0, lr1, delay_steps, delay_mult, max_steps) in enumerate(cases):
        f = gu.get_expon_lr_func(lr_init=lr0, lr_final=lr1, lr_delay_steps=delay_steps, lr_delay_mult=delay_mult,
                                 max_steps=max_steps)
        out[f"args_{i}"] = np.array([lr0, lr1, delay_steps, delay_mult, max_steps], np.float64)
        out[f"lr_{i}"] = np.array([float(f(int(s))) for s in steps], np.float64)
    np.savez(os.path.join(HERE, "lr_schedule.npz"), **out)
    print("wrote lr_schedule.npz")

    # the remaining losses of utils/loss_utils.py (l1_loss_with_conf :20-24, l2_loss :26-27, smooth_loss :36-44) and
    # image_utils.mse (:15-16) on seeded inputs
    import torch
    from utils import loss_utils as lu
    rng = np.random.default_rng(31)
    a = torch.tensor(rng.uniform(0, 1, (3, 37, 52)).astype(np.float32))
    b = torch.tensor(rng.uniform(0, 1, (3, 37, 52)).astype(np.float32))
    conf = torch.tensor(rng.uniform(0, 1, (1, 37, 52)).astype(np.float32))
    disp = torch.tensor(rng.uniform(0.1, 2, (1, 37, 52)).astype(np.float32))
    mse = (((a - b)) ** 2).view(a.shape[0], -1).mean(1, keepdim=True)  # image_utils.mse (that module imports matplotlib)
    np.savez(os.path.join(HERE, "losses_misc.npz"), a=a.numpy(), b=b.numpy(), conf=conf.numpy(), disp=disp.numpy(),
             l1_conf=lu.l1_loss_with_conf(a, b, conf).numpy(), l2=lu.l2_loss(a, b).numpy(),
             smooth=lu.smooth_loss(disp, a).numpy(), mse=mse.numpy())
    print("wrote losses_misc.npz")


if __name__ == "__main__":
    main()
