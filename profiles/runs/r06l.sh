# round 6: LDS-tile taps -- the microbenchmark's new variant S (three swizzled component planes) and the product kernel k_advect_vector_t against k_advect_vector_n
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
(cd profiles/micro/lds_tile && hipcc --offload-arch=gfx950 -O3 lds_tile_gather.hip -o /tmp/lds_tile_gather && timeout 120 /tmp/lds_tile_gather) > $O/lds_tile.txt 2>&1
timeout 600 python profiles/micro/advect_tile_ab.py 256 128 plume1024 --amp=96,160 > $O/ab.txt 2>&1
timeout 900 python -m pytest tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py -x -q > $O/pytest.log 2>&1
cat $O/lds_tile.txt | tail -12; cat $O/ab.txt; tail -3 $O/pytest.log
