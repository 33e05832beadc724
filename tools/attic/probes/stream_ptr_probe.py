import sys, timeit
sys.path.insert(0, "/root/repo")
import torch
from autoposeestimation_amd import _lib
torch.cuda.init()
a = _lib.stream_ptr().value
b = torch.cuda.current_stream().cuda_stream
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    c = _lib.stream_ptr().value
    assert c == s.cuda_stream, (c, s.cuda_stream)
assert (a or 0) == b, (a, b)
print("fast %.2f us  slow %.2f us" % (timeit.timeit(_lib.stream_ptr, number=20000) / 20000 * 1e6,
                                     timeit.timeit(lambda: torch.cuda.current_stream().cuda_stream, number=20000) / 20000 * 1e6))
