#!/usr/bin/env python3
"""bf16x6 kernels (fp32 operands as three bf16 pieces, six piece products) against their exact-fp32 twins
(dmp_dev_set_exact_fp32(1): v_mfma_f32_32x32x2_f32) and fp64 on adversarial operands.  Prints, per kernel and scenario, the
largest error of either form relative to the natural scale of the element, sum_k |a_k| |b_k|."""
import json
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bf16x6_cases import run_all
    print(json.dumps(run_all(th.device("cuda:0")), indent=1))


if __name__ == "__main__":
    main()
