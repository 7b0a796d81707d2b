#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/profile_sweeps.py > gpurun_out/sweeps.txt 2> gpurun_out/sweeps.err; echo rc=$?; cat gpurun_out/sweeps.txt; tail -3 gpurun_out/sweeps.err
