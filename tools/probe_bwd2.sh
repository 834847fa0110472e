#!/bin/bash
# bwd2 (roles on separate waves) against the two-launch backward and, bit for bit, against bwd1; interleaved timing
BWD_MODE=2 timeout 300 tools/probes/attn_bwd1_probe
