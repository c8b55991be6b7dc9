cd $GRAFT_REPO_ROOT
STEPS=4 WARMUP=2 timeout 900 bash tools/gpu_variants.sh edg2 edg1 edg2f4 edg2 edg1 edg2f4
