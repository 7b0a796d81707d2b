#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_end_to_end.py -m gpu -q -x 2>&1 | tail -5
