for wt in 0 8000 16000 24000 60000; do
  for fl in 0 450 1800; do
    echo "WT=$wt FLOOR=$fl" >> gpurun_out/r5_wt.txt
    if [ $wt -eq 0 ]; then unset DC_WAVE_TARGET; else export DC_WAVE_TARGET=$wt; fi
    if [ $fl -eq 0 ]; then unset DC_SHARE_FLOOR; else export DC_SHARE_FLOOR=$fl; fi
    python3 scratch/seg_bench.py 1000000 10 8 2>/dev/null | grep SEG >> gpurun_out/r5_wt.txt
  done
done
