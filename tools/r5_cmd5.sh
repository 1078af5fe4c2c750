cd $GRAFT_REPO_ROOT
g() { echo "PA_WG_GROUP_S9=$1 PA_WG_GROUP_S1=$2 $3"; }
bash tools/sweep_wq.sh "$(g 32 64)" "$(g 32 96)" "$(g 32 128)" "$(g 48 64)" "$(g 48 96)" "$(g 64 64)" "$(g 64 128)" "$(g 24 64)" "$(g 40 80)" "$(g 32 64 PA_WG_GROUP_MINPER9=2)" "$(g 32 64 PA_WG_GROUP_MINPER9=1)" "$(g 32 64 PA_WG_GROUP_MINPER1=1)" "$(g 32 64 PA_WG_GROUP_MINPER9=1\ PA_WG_GROUP_MINPER1=1)" "$(g 32 64)" > gpurun_out/sweep_wq3.txt 2>&1
