# usage: bash tools/gpu_r4a.sh <tag> -- member-mask backward: bit-identity + parity suites, then same-box A/B of bwd_masks
cd $GRAFT_REPO_ROOT
T=${1:-r4a}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_geometry_cache.py tests/test_gpu_fuzz.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -12 > gpurun_out/${T}_pytest.log
python tools/ab_variants.py bwd_masks 0 1 --bwd > gpurun_out/${T}_ab_stage.txt 2>&1
python tools/ab_step.py bwd_masks 0 1 > gpurun_out/${T}_ab_step.txt 2>&1
tail -5 gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_ab_stage.txt gpurun_out/${T}_ab_step.txt
