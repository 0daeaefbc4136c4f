mkdir -p gpurun_out/r04
python scripts/arx2_check.py --label spread > gpurun_out/r04/arx2_check2.txt 2>&1
tail -8 gpurun_out/r04/arx2_check2.txt | cut -c1-600
