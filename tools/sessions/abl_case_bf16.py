#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
tag = sys.argv[1] if len(sys.argv) > 1 else ""
sweep.bf16_case(4096, 1024, 64, 16, tag=tag + " C4 layer")
sweep.bf16_case(4096, 1024, 64, 128, tag=tag + " K=8192 (256 tiles)")
