# (variants/_pack_chain.so: `git show c91e169:polyfuzz_amd/csrc_host/_pack.c > /tmp/p.c && gcc -O3 -msse4.1 -shared -fPIC -I <python include> /tmp/p.c -lm -lpthread -o variants/_pack_chain.so`;
#  variants/_pack_pool.so: the shipped polyfuzz_amd/_pack.so)
# round 6: the crews' helpers started per call (a chain of pthread_create: variants/_pack_chain.so, built from the commit before) against
# the pool of parked threads (variants/_pack_pool.so = the shipped helper): TFIDF.match + the enqueue stamps, alternating on one box
cp polyfuzz_amd/_pack.so /tmp/_pack_keep.so
for rep in 1 2 3; do for v in chain pool; do
  cp variants/_pack_$v.so polyfuzz_amd/_pack.so
  echo "== $v"; python tools/r6_match_ab.py "s12:" "t4:PFZ_HOST_THREADS=4" | tail -2; python tools/r6_enqueue_stamps.py | tail -1
done; done
cp /tmp/_pack_keep.so polyfuzz_amd/_pack.so
