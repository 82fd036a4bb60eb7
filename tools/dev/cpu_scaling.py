"""DEV: how the reference's CPU path scales with the number of worker processes on this box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      "affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; cat /proc/loadavg")
for procs in (1, 4, 16, 32, 64, 128, 256):
    env = dict(os.environ, FREESASA_CPU_BASELINE_PROCS=str(procs))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--structs", str(max(procs, 8)), "--budget", "1.5"],
                         capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(procs, "procs:", "%.4g atoms/s" % d["value"], "ratio to procs x single %.2f" % d["all_cores_vs_cores_x_single"], "single %.3g" % d["single_thread"])
