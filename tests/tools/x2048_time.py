import sys
sys.argv=["bench.py"]
sys.path.insert(0, ".")
import bench
ss=bench.StressStep(0,0,1,None)
a=ss.time(50, False); b=ss.time(50, False)
print("x2048 ms/step", a, b, "solved", ss.solved_fraction())
ss.close()
