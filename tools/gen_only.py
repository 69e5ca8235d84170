"""Dev tool: bf16 resident generation only (for rocprofv3 traces)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gen_bench import run
run(sys.argv[1] if len(sys.argv) > 1 else "bf16", 3)
