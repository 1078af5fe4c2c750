cd $GRAFT_REPO_ROOT
g() { echo "PA_WG_GROUP_CAP=$1 PA_WG_GROUP_WGS9=64 PA_WG_GROUP_WGS1=64 PA_WG_GROUP_MINPER9=$2 PA_WG_GROUP_MINPER1=$3 $4"; }
bash tools/sweep_wq.sh "$(g 256 4 2)" "$(g 256 4 1)" "$(g 256 4 3)" "$(g 256 2 2)" "$(g 256 2 1)" "$(g 256 1 1)" "$(g 320 4 2)" "$(g 256 4 2 PA_WGRAD_NOPIPE=1)" "$(g 256 4 2 PA_STEM_ON_MAIN=1)" "$(g 256 4 2)" > gpurun_out/sweep_wq13.txt 2>&1
