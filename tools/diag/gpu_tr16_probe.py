import torch, sys
sys.path.insert(0,'.')
from torch_vae_amd import _lib
rc = _lib.lib().vae_selftest_tr16(torch.cuda.current_stream().cuda_stream)
print("rc", rc, _lib.lib().vae_last_error().decode())
