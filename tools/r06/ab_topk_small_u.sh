# A/B harness of round 6 (run on the GPU box through gpurun): environment switches of csrc/cdr_gemm.hip (CDR_TOPK_SEED, CDR_TOPK_SMALL_U).
cd $GRAFT_REPO_ROOT
for U in 128 256 512; do for S in 64 128 256 512; do echo "U=$U small_u=$S: $(CDR_TOPK_SMALL_U=$S MB_U=$U python tools/mb_fullsort_topk.py 128 2>/dev/null | tail -1)"; done; done
for U in 256; do for S in 64 256; do echo "D=64 U=$U small_u=$S: $(CDR_TOPK_SMALL_U=$S MB_U=$U python tools/mb_fullsort_topk.py 64 2>/dev/null | tail -1)"; done; done
