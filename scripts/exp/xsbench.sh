# usage: xsbench.sh "<lib-variant or ->:<nb> ..."
for item in $1; do lib=${item%%:*}; nb=${item##*:}
  if [ "$lib" != "-" ]; then export OSC_LIB_PATH=$PWD/oscillink_amd/liboscillink_hip_$lib.so; else unset OSC_LIB_PATH; fi
  OSC_SPMM_XS=1 OSC_XS_NB=$nb python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $nb, round(d['ms_per_step'],3), round(d['roofline']['apply_ms'],4))"
done
