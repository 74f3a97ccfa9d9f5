#!/bin/bash
# re-tune of the stream-assignment switches on the final tree: each switch's alternatives against its default, alternating, one box
cd $GRAFT_REPO_ROOT
bash tools/r6_ab3.sh ECHR_DXT_STREAM "2 1 0" 2
bash tools/r6_ab3.sh ECHR_TSRM_FORK2 "0 1" 2
bash tools/r6_ab3.sh ECHR_FORK_FIRST "1 0" 2
bash tools/r6_ab3.sh ECHR_EVB0_FIRST "1 0" 2
bash tools/r6_ab3.sh ECHR_WFC1_ON_CALLER "1 0" 2
bash tools/r6_ab3.sh ECHR_ADAM_NT "2 3 1" 2
bash tools/r6_ab3.sh ECHR_PERSIST_PREBUILD "1 0" 2
bash tools/r6_ab3.sh ECHR_ASYNC_LEVEL "2 1" 2
