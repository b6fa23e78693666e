for rep in 1 2; do
DLSCQP_LIB=$PWD/dlsc_gc_planner_amd/csrc/libdlscqp_glds.so python3 tests/tools/crowd_time.py 2>&1 | tail -1
python3 tests/tools/crowd_time.py 2>&1 | tail -1
done
DLSCQP_LIB=$PWD/dlsc_gc_planner_amd/csrc/libdlscqp_glds.so python3 tests/tools/small_time.py 2>&1 | tail -6
python3 tests/tools/small_time.py 2>&1 | tail -6
