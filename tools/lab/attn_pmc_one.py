#!/usr/bin/env python3
"""Lab: a few launches of the bench attention shape (B=256, L=256, rel-key, default arithmetic) for rocprofv3 --pmc passes."""
import os, sys
sys.argv = [sys.argv[0], "attn_pmc"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_kernels
bench_kernels.attn_pmc()
