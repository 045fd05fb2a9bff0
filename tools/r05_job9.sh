#!/bin/bash
# r05 GPU job 9: (a) build() from CLEAN on the GPU box + smoke; (b) the vendor GEMM library's rate on the encoder's shapes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
make clean > /dev/null 2>&1
( time python __graft_entry__.py smoke ) > gpurun_out/job9_clean_build.log 2>&1; echo "clean build + smoke rc=$?"; tail -6 gpurun_out/job9_clean_build.log
cp gpurun_out/build_record.json gpurun_out/job9_build_record_clean.json
python tools/ubench/lib_gemm_rate.py > gpurun_out/job9_lib_gemm.log 2>&1; tail -1 gpurun_out/job9_lib_gemm.log
TORCH_BLAS_PREFER_HIPBLASLT=1 python tools/ubench/lib_gemm_rate.py >> gpurun_out/job9_lib_gemm.log 2>&1; tail -1 gpurun_out/job9_lib_gemm.log
TORCH_BLAS_PREFER_HIPBLASLT=0 python tools/ubench/lib_gemm_rate.py >> gpurun_out/job9_lib_gemm.log 2>&1; tail -1 gpurun_out/job9_lib_gemm.log
python tools/t5_bench.py 8 1200 > gpurun_out/job9_t5_bench.log 2>&1; tail -2 gpurun_out/job9_t5_bench.log
UC_TIMING=1 timeout 900 python tools/workflow_at_size.py 2000 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job9_line_nominal.json 2> gpurun_out/job9_timing_nominal.log; echo "nominal rc=$?"
tail -c 600 gpurun_out/job9_line_nominal.json
