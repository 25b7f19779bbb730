#!/usr/bin/env python3
"""times the f32 BRGEMM for the library named by TPP_XSMM_LIBRARY (ablation builds)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep  # noqa: E402  (reuses time_it / f32_case; its __main__ block does not run)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
forces = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for v in forces:
    sweep.f32_case(1024, 1024, 64, 16, force=v, tag=tag + " C2")
    sweep.f32_case(1024, 1024, 64, 128, force=v, tag=tag + " K=8192")
