set -e
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "deblock" > gpurun_out/ab/parity_db.log 2>&1 || { tail -40 gpurun_out/ab/parity_db.log; exit 1; }
tail -2 gpurun_out/ab/parity_db.log
for r in 4 8; do
echo "rows $r base: $(MI355ENC_DB_ROWS=$r timeout -k 10 120 python tools/probe_deblock_real.py)"
done
for r in 4 8; do echo "rows $r"; MI355ENC_DB_ROWS=$r MI355ENC_LIB=$PWD/ceracoder_amd/variants/libmi355enc_PROF.so timeout -k 10 120 python tests/devtools/dbrprof.py; done
