#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/r06_s.sh
bash tools/r06_p.sh
