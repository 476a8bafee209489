import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from yolo_deepsort_amd import _lib
_lib.init(); lib=_lib.load()
print("rccl", lib.yds_comm_rccl_version())
ident=(C.c_char*128)()
print("uid rc", lib.yds_comm_unique_id(ident), _lib.last_error() )
p=lib.yds_comm_create(ident,1,0)
print("comm", p, _lib.last_error())
