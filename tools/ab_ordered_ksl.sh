cd recbole-cdr_amd/csrc
mkdir -p /tmp/ab
for v in "4 1024" "2 1024" "1 1024" "2 512" "1 512"; do
  set -- $v
  sed "s/constexpr int kSL = [0-9];/constexpr int kSL = $1;/; s/constexpr int kBlock = 1024;/constexpr int kBlock = $2;/" cdr_ordered.hip > /tmp/ab/ord_$1_$2.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I. -I../../include -c /tmp/ab/ord_$1_$2.hip -o /tmp/ab/ord_$1_$2.o || exit 1
  objs=$(ls build/*.o | grep -v cdr_ordered.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab/lib_$1_$2.so $objs /tmp/ab/ord_$1_$2.o || exit 1
done
cd ../..
for v in 4_1024 2_1024 1_1024 2_512 1_512; do echo "== kSL_kBlock $v"; CDR_LIB_PATH=/tmp/ab/lib_$v.so MB_QUICK=1 python tools/mb_ordered_bwd.py 2>/dev/null | tail -8 | cut -c40-170; done
