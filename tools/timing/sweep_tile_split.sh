for shp in "2048 768 768" "2048 2048 2048" "2048 4096 4096" "2048 768 3072"; do
  python3 tools/timing/sweep_tile_split.py $shp 2>&1 | tail -1
  for tr in 128 256; do for sp in 1 2 4 8; do
    MI355Q_V8_TILE_ROWS=$tr MI355Q_V8_SPLITS=$sp python3 tools/timing/sweep_tile_split.py $shp 2>&1 | tail -1
  done; done
done
