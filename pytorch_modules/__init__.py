"""Import shim for the reference's external block library (github.com/WoodsGao/pytorch_modules, not vendored in the
reference tree): the names the reference's scripts and model files import from it, bound to the MI355X-native
implementations.  Only what the hot path uses exists; the rest raises with a pointer."""
