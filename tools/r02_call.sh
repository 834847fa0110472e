#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash tools/gpu_round.sh r02z 40
