set -e
mkdir -p gpurun_out
python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_base.json 2> gpurun_out/r6_ml32_base.err
echo base done
DMM_ML_WS_CAP_MIB=8192 python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_ws8g.json 2> gpurun_out/r6_ml32_ws8g.err
echo 8g done
DMM_ML_WS_CAP_MIB=4096 python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_ws4g.json 2> gpurun_out/r6_ml32_ws4g.err
echo 4g done
DMM_ML_WS_CAP_MIB=2048 python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_ws2g.json 2> gpurun_out/r6_ml32_ws2g.err
echo 2g done
