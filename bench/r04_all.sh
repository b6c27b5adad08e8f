#!/bin/bash
# One GPU call for everything round 4 still owes: first contact + tests + bench line (r04_batch2.sh), then the profiles
# (collect_r04.sh).  The second half only runs when the first did not stop at the hub smoke test.
cd "$(dirname "$0")/.."
bash bench/r04_batch2.sh || exit 1
bash bench/collect_r04.sh
