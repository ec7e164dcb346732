"""xmca_amd - MI355X (gfx950) implementation of the xmca solve / rotate / rule_n path.

    from xmca_amd.array import MCA        # drop-in for xmca.array.MCA
    from xmca_amd.xarray import xMCA      # drop-in for xmca.xarray.xMCA (needs xarray)

The numerical core lives in libxmca_hip.so (HIP kernels + C ABI, see include/xmca_hip.h); build it with
`python -m xmca_amd.build`.
"""
__version__ = '0.2.0'
__all__ = ['__version__']
