#!/bin/bash
BWD_MODE=2 timeout 300 tools/probes/attn_bwd1_probe_prof | grep -A9 "median\|PROBE\|MISM\|cycles per item\|bwd2 vs" | grep -v "dq:\|dk:\|dv:\|dtab\|rerun" | head -50
