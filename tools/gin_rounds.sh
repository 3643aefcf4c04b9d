#!/bin/bash
# gradInput's gather kernel against the frame height: 768 workgroups are resident at a time
# (3 per CU), so 756 rows x 1920 are one full round, 1080 rows 1.42 rounds, 1520 rows two.
set -o pipefail
for h in 540 756 1080 1300 1520; do
  echo "== $h x 1920"
  timeout -k 10 200 python tools/sweep.py --shape 1,$h,1920 --radii=-3 --gin --sigmas 0.045 2>&1 | grep "gin atomic" | cut -c1-130 || exit 1
done
