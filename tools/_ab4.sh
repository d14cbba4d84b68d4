set -e
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "deblock" > gpurun_out/ab/parity_db.log 2>&1 || { tail -30 gpurun_out/ab/parity_db.log; exit 1; }
tail -2 gpurun_out/ab/parity_db.log
echo "new : $(timeout -k 10 120 python tools/probe_deblock_real.py)"
echo "HEAD: $(MI355ENC_LIB=$PWD/ceracoder_amd/variants/libmi355enc_HEAD.so timeout -k 10 120 python tools/probe_deblock_real.py)"
echo "new : $(timeout -k 10 120 python tools/probe_deblock_real.py)"
