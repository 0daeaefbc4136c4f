mkdir -p gpurun_out/r04
python scripts/arx3_check.py --label v3 > gpurun_out/r04/arx3_check.txt 2>&1
tail -32 gpurun_out/r04/arx3_check.txt | cut -c1-700
