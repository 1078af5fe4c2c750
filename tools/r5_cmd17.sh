cd $GRAFT_REPO_ROOT
g() { echo "PA_WG_GROUP_CAP=$1 PA_WG_GROUP_WGS9=$2 PA_WG_GROUP_WGS1=$3 $4"; }
bash tools/sweep_wq.sh "$(g 256 64 64)" "$(g 384 64 64)" "$(g 100000 64 64)" "$(g 256 48 64)" "$(g 256 64 48)" "$(g 256 48 48)" "$(g 256 80 80)" "$(g 192 64 64)" "$(g 256 64 64 PA_WG_GROUP_MINPER9=2)" "$(g 256 64 64 PA_WG_GROUP_MINPER1=2)" "$(g 256 64 64)" > gpurun_out/sweep_wq12.txt 2>&1
