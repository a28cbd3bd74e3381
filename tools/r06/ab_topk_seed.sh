# A/B harness of round 6 (run on the GPU box through gpurun): environment switches of csrc/cdr_gemm.hip (CDR_TOPK_SEED, CDR_TOPK_SMALL_U).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "topk or fullsort or evaluate or full_c5" 2>&1 | tail -4
for U in 64 128 256 512 1024; do for S in 0 1; do echo "U=$U seed=$S: $(CDR_TOPK_SEED=$S MB_U=$U python tools/mb_fullsort_topk.py 128 2>/dev/null | tail -1)"; done; done
for U in 256 1024; do for S in 0 1; do echo "D=64 U=$U seed=$S: $(CDR_TOPK_SEED=$S MB_U=$U python tools/mb_fullsort_topk.py 64 2>/dev/null | tail -1)"; done; done
