#!/usr/bin/env python3
"""torch.profiler view of one optimisation iteration (tools/bench_iteration.py): top device-time ops."""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench_iteration as B  # noqa

if __name__ == "__main__":
    sys.argv = [sys.argv[0], "--iters", "2"]
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        B.main()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
