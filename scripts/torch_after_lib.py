"""Order experiment: HIP initialised by our library first, torch.cuda afterwards."""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CASES = {
 "stream_then_torch": "import cphnsw_mi355x as c; s=c.FastScanStream(128,4,300); s.run(1); s.close(); import torch; torch.cuda.init(); print('ok', torch.cuda.device_count())",
 "stream2048_then_torch": "import cphnsw_mi355x as c; s=c.FastScanStream(2048,4,300); s.run(1); s.close(); import torch; torch.cuda.init(); print('ok', torch.cuda.device_count())",
 "index_then_torch": "import cphnsw_mi355x as c; i=c.CPIndex(128,4); import torch; torch.cuda.init(); print('ok', torch.cuda.device_count())",
 "torchimport_stream_torchinit": "import torch; import cphnsw_mi355x as c; s=c.FastScanStream(128,4,300); s.run(1); s.close(); torch.cuda.init(); print('ok', torch.cuda.device_count())",
 "torchimport_index_torchinit": "import torch; import cphnsw_mi355x as c; i=c.CPIndex(128,4); torch.cuda.init(); print('ok', torch.cuda.device_count())",
 "ldd": "import subprocess, cphnsw_mi355x._lib as l, cphnsw_mi355x.build as b; print(subprocess.run(['ldd', b.LIB_PATH],capture_output=True,text=True).stdout)",
 "maps": "import torch; import cphnsw_mi355x as c; i=c.CPIndex(128,4); print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l][::8])",
}
for name, code in CASES.items():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); " % os.path.join(ROOT, "rabitq-ann-search_amd") + code], capture_output=True, text=True)
    print("==", name, "rc", r.returncode, (r.stdout.strip() or r.stderr.strip()[-300:]))
