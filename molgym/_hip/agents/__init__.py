import molgym as _m

__path__ = _m.search_path('agents')
