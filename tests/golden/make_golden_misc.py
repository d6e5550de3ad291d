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


if __name__ == "__main__":
    main()
