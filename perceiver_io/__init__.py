"""Import-path shim: the reference's example scripts do `from perceiver_io.<module> import <Class>`.  Every name
resolves to the MI355X-native implementation in `perceiverio_pytorch_amd`."""
