#!/bin/bash
set -u
bash tools/round3/ab.sh base
bash tools/round3/ab.sh sel_no_pair_stores KAMD_SELECT_MODE=1
bash tools/round3/ab.sh sel_no_pair_loop KAMD_SELECT_MODE=2
bash tools/round3/ab.sh sel_no_chunks KAMD_SELECT_MODE=4
