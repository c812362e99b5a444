#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r05/s11.sh
bash tools/r05/s9.sh
