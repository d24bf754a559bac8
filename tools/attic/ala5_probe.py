"""Development aid: the ala5-shaped forces series of bench.py (ala5_record) on its own."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd, bench
print(json.dumps(bench.ala5_record(bioen_amd, bench.SEED), indent=1))
