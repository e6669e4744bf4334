import sys, time
sys.path.insert(0,'.')
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle","synth"])
files=[H.synth_jpeg(width=1920,height=1080,seed=100+i) for i in range(16)]
for n in (16,256):
    b=J.JpegBatch()
    for f in files: b.add_jpeg(f)
    b.tile(n)
    t=time.time(); b.upload(); print('upload %.1f ms'%((time.time()-t)*1e3))
    b.decode(); b.sync()
    ms,st=b.decode_timed(3)
    print(n,'images: %.2f ms/decode'%ms, {k:round(v,3) for k,v in st.items()}, 'Mpix/s=%.0f'%(b.pixels()/ms/1e3), 'algGB/s=%.1f'%(b.algorithmic_bytes()/ms/1e6))
    b.close()
