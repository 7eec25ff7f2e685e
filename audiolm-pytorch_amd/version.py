__version__ = '2.4.0'   # mirrors the reference version this drop-in tracks (/root/reference/audiolm_pytorch/version.py)
