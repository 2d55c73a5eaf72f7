#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh r03 2>&1 | tail -20
bash tools/collect_ablations.sh r03 2>&1 | tail -3
