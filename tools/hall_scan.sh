for kv in "" "WV_PAIR_UNIT_PLANES=16" "WV_PAIR_UNIT_PLANES=24" "WV_PAIR_UNIT_PLANES=48" "WV_PAIR_UNIT_PLANES=64" "WV_PAIR_UNITS_BY_CHUNK=0" "WV_PAIR_UNIT_WAVES=0" "WV_TILE_LISTS=0" "WV_PAIR_CHUNKS=8" "WV_PAIR_CHUNKS=12"; do
  echo "== $kv"; env $kv python tools/concert_bench.py 800 2>&1 | grep -E "engine's choice|two-step passes  "
done
