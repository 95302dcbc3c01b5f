export PERF_WARM=20 PERF_N=40
python tools/_memprobe.py 2>/dev/null
for i in 1 2 3; do
  for x in np 0; do
    if [ $x = 0 ]; then L=acids_transforms_amd/libacids_hip.so; else L=tools/ab/libacids_$x.so; fi
    echo "== $x round $i"; ACIDS_HIP_LIB=$PWD/$L python tools/perf_all.py sizes 2>/dev/null | grep -i "STFT(4096\|STFT(512\|STFT(2048\|MFCC(" | cut -c1-60 | tr -s ' ' | tr '\n' '|'; echo
  done
done
