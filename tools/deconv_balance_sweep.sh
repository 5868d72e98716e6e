timeout 600 python -m pytest tests/test_conv_multi_gpu.py -x -q -m gpu 2>&1 | tail -2
for e in 0 6 10 16 24; do echo -n "SDF_DECONV_EPI=$e: "; SDF_DECONV_EPI=$e bash tools/prof_forward_one.sh r5av > /dev/null 2>&1; grep "spike_deconv_wres_kernel" gpurun_out/prof_r5av_sequence.txt | tail -1 | cut -c1-90; done
echo -n "balance off: "; SDF_DECONV_BALANCE=0 bash tools/prof_forward_one.sh r5av > /dev/null 2>&1; grep "spike_deconv_wres_kernel" gpurun_out/prof_r5av_sequence.txt | tail -1 | cut -c1-90
