"""Drop-in package `model` for AutoFormer: provides `model.module.*` and `model.utils`
(the import paths of AutoFormer/model/supernet_transformer.py:6-11) backed by
cream_amd.autoformer.  `supernet_transformer.py` itself is NOT provided here: the
reference's own file is found through the extended package path (see
cream_amd.dropin.install_autoformer), i.e. it runs unchanged on top of these modules."""
