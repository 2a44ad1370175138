import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import fused, ops
for kv in filter(None, os.environ.get("DBG", "").split(",")):
    k, v = kv.split("=")
    mod = ops if hasattr(ops, k) and not hasattr(fused, k) else fused
    setattr(mod, k, bool(int(v)))
    print("set", mod.__name__, k, getattr(mod, k))
import pytest
sys.exit(pytest.main(sys.argv[1:]))
