#!/bin/bash
timeout 900 python -m pytest tests/test_comm_gpu.py -q -s 2>&1 | grep -v "^$" | tail -25 | cut -c1-300
