#!/bin/bash
# What bounds the fused physics rollouts: builds three measurement variants of the library into tools/ab/ (git-ignored,
# carried to the GPU box) in which small_obs_regs_rollout leaves out its observation rows (1), its reward / discount /
# step_type stores (2) or both (3), its action loads (4: the only reads inside the loop; 7: no memory traffic in the loop at all) — the arithmetic stays — and prints the gpurun command that times them next to the
# product library in ONE call (profiles/r03/exp_store_ablation.log).
set -eu
cd "$(git rev-parse --show-toplevel)"
python -m bsuite_amd.build | tail -1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Iinclude -Ibsuite_amd/csrc"
mkdir -p tools/ab
for e in ${ABLATE:-1 2 3}; do
  for f in cartpole mountain_car; do hipcc $FLAGS -DBSX_ABLATE_STORES=$e -c bsuite_amd/csrc/$f.hip -o /tmp/${f}_ablate$e.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libbsx_ablate$e.so /tmp/cartpole_ablate$e.o /tmp/mountain_car_ablate$e.o $(ls bsuite_amd/_lib/*.o | grep -v "/cartpole.o\|/mountain_car.o")
done
ls -la tools/ab/libbsx_ablate*.so
echo "gpurun -- 'for lib in \"\" tools/ab/libbsx_ablate1.so tools/ab/libbsx_ablate2.so tools/ab/libbsx_ablate3.so; do BSX_NATIVE_LIB=\$lib python tools/lanes_sweep.py --mode rollout --T 16 --steps 320 cartpole mountain_car -- 2**20; done'"
