import torch, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.smoke()
print("smoke under torch-first import order ok; torch", torch.__version__)
