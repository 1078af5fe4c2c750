cd $GRAFT_REPO_ROOT
g() { echo "PA_WG_GROUP_S9=$1 PA_WG_GROUP_S1=$2 $3"; }
bash tools/sweep_wq.sh "$(g 32 96)" "$(g 32 96 PA_WG_GROUP_PIPE=0)" "$(g 64 128 PA_WG_GROUP_PIPE=0)" "$(g 32 64 PA_WG_GROUP_PIPE=0)" "$(g 48 96 PA_WG_GROUP_PIPE=0)" "$(g 32 96 PA_WGRAD_NOPIPE=1)" "$(g 32 96 PA_WGRAD_ANYORDER=0)" "$(g 32 80)" "$(g 28 96)" "$(g 36 96)" "$(g 32 96 PA_WG_GROUP_MINPER1=1)" "$(g 32 96)" > gpurun_out/sweep_wq4.txt 2>&1
