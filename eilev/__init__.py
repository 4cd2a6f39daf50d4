"""Import-path alias: lets the reference's callers (`from eilev.model.v2 import ...`, `from eilev.data.utils import ...`)
run unchanged against the MI355X-native implementation in `eilev_amd`."""
