import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests"); sys.path.insert(0, %r + "/tools")
import common, oracle_py
from speedseq_amd import capi
lib = capi.Lib(); orc = oracle_py.Oracle(%r + "/oracle/liboracle.so")
prefix = common.repeat_reference(orc, "/tmp/rep")
print("regions", common.check_align1(lib, orc, 300, seed=22, prefix=prefix))
''' % (ROOT, ROOT, ROOT, ROOT)
os.makedirs("/tmp/rep", exist_ok=True)
for env in ({"SSG_CHAIN_CAP_TEST": "40", "SSG_CHAIN_CAP_TEST_REDO": "0", "SSG_DEBUG": "1", "SSG_CHAIN_WAVE_MIN": "100"},):
    e = dict(os.environ, **env)
    try:
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=35)
        print(env, "rc", r.returncode, r.stdout[-200:], r.stderr[-600:])
    except subprocess.TimeoutExpired as t:
        print(env, "TIMEOUT", (t.stderr or b"")[-900:])
