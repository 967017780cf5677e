"""Import-path alias: the module paths `train/train_artiboost.py:9-22` of lixiny/ArtiBoost imports, resolved to the
MI355X-native build in `artiboost_amd` -- so that script (and configs naming the registry TYPEs) run without an edit.
Every module here is a re-export; the implementations, and their reference citations, live in `artiboost_amd/`."""
