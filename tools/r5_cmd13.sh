cd $GRAFT_REPO_ROOT
g() { echo "PA_WG_GROUP_S9=$1 PA_WG_GROUP_S1=$2 $3"; }
bash tools/sweep_wq.sh "$(g 24 64)" "$(g 24 48)" "$(g 16 48)" "$(g 16 64)" "$(g 20 56)" "$(g 32 48)" "$(g 24 64 PA_STEM_ON_MAIN=1)" "$(g 24 64 PA_WG_GROUP_MINPER1=4)" "$(g 24 64 PA_WG_GROUP_MINPER9=8)" "$(g 24 64)" "$(g 24 64 PA_STEM_ON_MAIN=1)" > gpurun_out/sweep_wq8.txt 2>&1
