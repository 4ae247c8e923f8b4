"""Data path of the co-training runs (reference: generalframework/dataset/)."""
from .medicalDataLoader import MedicalImageDataset, DeviceSliceCache, CachedLoader  # noqa: F401
from .augment import segment_transform, ToLabel, PILaugment  # noqa: F401
from .ACDC_helper import (PatientSampler, get_ACDC_dataloaders, get_ACDC_split_dataloders, extract_patients,  # noqa: F401
                          create_partitions, to_cached_loaders)
from .GM_helper import get_GM_dataloaders, get_GMC_split_dataloders, extract_patients_gmc  # noqa: F401
