#!/bin/bash
BWD_MODE=2 timeout 300 tools/probes/attn_bwd1_probe_prof | grep "median\|PROBE\|MISM"
