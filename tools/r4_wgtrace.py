"""Print the shape of every weight-gradient launch of one training step (TRAJSDE_WGRAD_TRACE) and time each with HIP events."""
import os, sys
os.environ["TRAJSDE_WGRAD_TRACE"] = "1"
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "0"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_step_bench
train_step_bench.main()
