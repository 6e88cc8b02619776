import sys
sys.path.insert(0,'.')
import hwang
from tests.corpus_util import make_clip
make_clip('/tmp/a.mp4', yuv=False, frames=24, gop=12, bframes=2)
for nd in (1,4):
    h=hwang.DeviceHandle(); h.type=hwang.DeviceType.GPU; h.id=0
    print('handle', h.type, h.id, flush=True)
    d=hwang.Decoder('/tmp/a.mp4', device_type=hwang.DeviceType.GPU, num_devices=nd)
    fr=d.retrieve([0,3,13])
    print(nd, len(fr), fr[0].shape, flush=True)
