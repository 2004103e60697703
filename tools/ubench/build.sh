#!/bin/bash
# builds the stand-alone microbenchmarks next to their sources (tools/ubench/*.bin); run them on the GPU box
cd "$(dirname "$0")"
hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate.bin -w
hipcc -O3 --offload-arch=gfx950 -std=c++17 -w igemm_bench.cpp -o igemm_bench.bin -L../../bndm_amd -lbndm_hip -Wl,-rpath,'$ORIGIN/../../bndm_amd'
hipcc -O3 --offload-arch=gfx950 -std=c++17 -w noise_bench.cpp -o noise_bench.bin -L../../bndm_amd -lbndm_hip -Wl,-rpath,'$ORIGIN/../../bndm_amd'
hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tail_bench.cpp -o tail_bench.bin -L../../bndm_amd -lbndm_hip -Wl,-rpath,'$ORIGIN/../../bndm_amd'
hipcc -O3 --offload-arch=gfx950 -w covalu.hip -o covalu.bin
hipcc -O3 --offload-arch=gfx950 -w hwid.hip -o hwid.bin
hipcc -O3 --offload-arch=gfx950 -w coldload.hip -o coldload.bin
hipcc -O3 --offload-arch=gfx950 -std=c++17 -w t32_bench.cpp -o t32_bench.bin -L../../bndm_amd -lbndm_hip -Wl,-rpath,'$ORIGIN/../../bndm_amd'
hipcc -O3 --offload-arch=gfx950 -w anyorder.hip -o anyorder.bin
hipcc -O3 --offload-arch=gfx950 -w flagwait.hip -o flagwait.bin
hipcc -O3 --offload-arch=gfx950 -w cumask_probe.hip -o cumask_probe.bin
