import os, sys, pytest
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")

# A lone process spans every device it sees (mm_align_init).  The suite is written for, and has only ever run on, one-GPU boxes: on a node with several GPUs every command-line
# run below would otherwise span them all -- a configuration with a test of its own (test_multi_gpu: MM_DEVICE_CONTEXTS), not one to meet by accident.  One device unless a test says otherwise.
os.environ.setdefault('MM_DEVICES', '1')
