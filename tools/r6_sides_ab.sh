# round 6: the streamed session's range chains on 1 / 2 / 3 / 4 side streams (variants built by tools/build_variant.sh
# variants/sidesK.so -DPFZ_K3_SYM_SIDE_STREAMS=K; the shipped library is the default), TFIDF.match(names) on one box, alternating
for rep in 1 2; do
for lib in "" variants/sides1.so variants/sides3.so variants/sides4.so; do
  echo "== lib=${lib:-default}"
  POLYFUZZ_HIP_LIB=${lib:+$PWD/$lib} python tools/r6_match_ab.py "s12:" "s16:PFZ_MATCH_SHARES=1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1" "s24:PFZ_MATCH_SHARES=1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1"
done; done
for lib in variants/sides3.so variants/sides4.so; do POLYFUZZ_HIP_LIB=$PWD/$lib python tools/r6_streamed_soak.py 200 2>&1 | tail -1; done
