python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "surface_scene_matches_reference_fixture" 2>&1 | grep -v "Warn\|warn" | grep -v "^  ray " > gpurun_out/r4_t6a.log
NEUMESH_MLP_PRECISION=f16x2 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "surface_scene_matches_reference_fixture" 2>&1 | grep -v "Warn\|warn" | grep -v "^  ray " | tail -30 > gpurun_out/r4_t6b.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -k "trainer_step or training_render or painting_step or train" 2>&1 | grep -v "Warn\|warn" | tail -30 > gpurun_out/r4_t6c.log
python tools/train_profile.py > gpurun_out/r4_train_profile.log 2>&1
NEUMESH_STAGED_SAMPLER=1 python tools/train_profile.py > gpurun_out/r4_train_profile_staged.log 2>&1
tail -3 gpurun_out/r4_t6a.log gpurun_out/r4_t6c.log
