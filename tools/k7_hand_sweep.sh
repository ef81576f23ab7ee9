for h in "12,16,16" "24,16,8" "32,16,8" "48,16,8" "32,8,16" "64,16,16" "9999,16,8"; do
  echo "== PFZ_K7_HAND=$h"; PFZ_K7_HAND=$h timeout 100 python tools/k7_time.py 20000 WRatio,token_ratio,partial_ratio 2>&1 | grep -E "^(WRatio|token_ratio|partial_ratio)" | cut -c1-110
done
