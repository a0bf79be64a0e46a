"""Host-side mirrors of the reference's ``models`` package (sslassist.AModel, xlsr.SSLModel, senet).
Unlike the reference's models/__init__.py:1-3 importing this package pulls nothing heavy."""
