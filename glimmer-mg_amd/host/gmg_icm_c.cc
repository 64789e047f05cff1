//  gmg_icm_c.cc -- extern "C" wrappers of include/gmg_icm.h over the host ICM_t.

#include "icm.hh"
#include "../../include/gmg_icm.h"
#include "../csrc/gmg_internal.h"   // gmg_set_error

#include <errno.h>
#include <string.h>
#include <new>

using namespace std;

//  The reference keeps this global in src/Common/delcher.cc:20; an application
//  that links its own definition overrides this one.
__attribute__ ((weak)) int  Verbose = 0;

struct gmg_icm
  {
   ICM_Training_t  model;     // an ICM_t with Train_Model; no extra state
   gmg_icm  (int w, int d, int p) : model (w, d, p) {}
  };

extern "C" int  gmg_icm_new  (int w, int d, int p, gmg_icm * * out)
  {
   if  (out == NULL || w < 1 || d < 0 || d > 12 || p < 1)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_new: bad argument");
   * out = new (nothrow) gmg_icm (w, d, p);
   if  (* out == NULL)
       return  gmg_set_error (GMG_ENOMEM, "gmg_icm_new: out of memory");
   return  GMG_OK;
  }

extern "C" int  gmg_icm_train
    (const char * const * strings, int n_strings, int w, int d, int p, gmg_icm * * out)
  {
   if  (out == NULL || (strings == NULL && n_strings > 0) || n_strings < 0 || w < 1 || d < 0 || d > 12 || p < 1)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_train: bad argument");
   gmg_icm  * h = new (nothrow) gmg_icm (w, d, p);
   if  (h == NULL)
       return  gmg_set_error (GMG_ENOMEM, "gmg_icm_train: out of memory");
   string  err;
   if  (! h -> model . Try_Train_Model (strings, n_strings, err))
       {
        delete  h;
        return  gmg_set_error (GMG_EHIP, "%s", err . c_str ());
       }
   * out = h;
   return  GMG_OK;
  }

extern "C" int  gmg_icm_open  (const char * path, gmg_icm * * out)
  {
   if  (path == NULL || out == NULL)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_open: NULL argument");
   FILE  * fp = fopen (path, "rb");
   if  (fp == NULL)
       return  gmg_set_error (GMG_EINVAL, "ERROR:  Could not open file  %s  errno = %d", path, errno);
   gmg_icm  * h = new (nothrow) gmg_icm (1, 0, 1);
   if  (h == NULL)
       {
        fclose (fp);
        return  gmg_set_error (GMG_ENOMEM, "gmg_icm_open: out of memory");
       }
   string  err;
   bool  ok = h -> model . Try_Input (fp, err);
   fclose (fp);
   if  (! ok)
       {
        delete  h;
        return  gmg_set_error (GMG_EBADMODEL, "%s", err . c_str ());
       }
   * out = h;
   return  GMG_OK;
  }

extern "C" int  gmg_icm_build_indep
    (gmg_icm * icm, double gc_frac, const char * const * stop_codon, int n_stops)
  {
   if  (icm == NULL || (n_stops > 0 && stop_codon == NULL))
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_build_indep: NULL argument");
   ICM_t  & m = icm -> model;
   if  (m . Get_Model_Len () != 3 || m . Get_Model_Depth () != 2 || m . Get_Periodicity () != 3
          || m . Get_Num_Nodes () != 21)
       return  gmg_set_error (GMG_EBADMODEL, "ERROR:  Incompatible ICM_Training_t for Build_Indep_WO_Stops");
   vector <const char *>  stops;
   for  (int i = 0;  i < n_stops;  i ++)
     {
      if  (stop_codon [i] == NULL || strlen (stop_codon [i]) < 3)
          return  gmg_set_error (GMG_EINVAL, "gmg_icm_build_indep: stop codon %d is not 3 letters", i);
      stops . push_back (stop_codon [i]);
     }
   m . Build_Indep_WO_Stops (gc_frac, stops);
   return  GMG_OK;
  }

extern "C" int  gmg_icm_write  (gmg_icm * icm, const char * path)
  {
   if  (icm == NULL || path == NULL)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_write: NULL argument");
   FILE  * fp = fopen (path, "wb");
   if  (fp == NULL)
       return  gmg_set_error (GMG_EINVAL, "ERROR:  Could not open file  %s  errno = %d", path, errno);
   icm -> model . Output (fp, true);
   if  (fclose (fp) != 0)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_write: write to %s failed", path);
   return  GMG_OK;
  }

extern "C" int  gmg_icm_free  (gmg_icm * icm)
  {
   delete  icm;
   return  GMG_OK;
  }

extern "C" int  gmg_icm_params
    (const gmg_icm * icm, int * w, int * d, int * p, int * n)
  {
   if  (icm == NULL)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_params: NULL model");
   ICM_t  & m = const_cast <ICM_Training_t &> (icm -> model);
   if  (w)  * w = m . Get_Model_Len ();
   if  (d)  * d = m . Get_Model_Depth ();
   if  (p)  * p = m . Get_Periodicity ();
   if  (n)  * n = m . Get_Num_Nodes ();
   return  GMG_OK;
  }

extern "C" int  gmg_icm_tables  (const gmg_icm * icm, int16_t * mip, float * prob4)
  {
   if  (icm == NULL || mip == NULL || prob4 == NULL)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_tables: NULL argument");
   vector <short>  m;
   vector <float>  p;
   icm -> model . Export_Tables (m, p);
   memcpy (mip, m . data (), m . size () * sizeof (short));
   memcpy (prob4, p . data (), p . size () * sizeof (float));
   return  GMG_OK;
  }

extern "C" int  gmg_icm_device_model  (const gmg_icm * icm, const gmg_model * * out)
  {
   if  (icm == NULL || out == NULL)
       return  gmg_set_error (GMG_EINVAL, "gmg_icm_device_model: NULL argument");
   //  ICM_t::Device_Model exits on failure (reference convention); check the one likely cause first
   //  so that this C entry point returns a status instead
   if  (gmg_device_count () <= 0)
       return  gmg_set_error (GMG_ENODEV, "gmg_icm_device_model: no HIP device; there is no CPU fallback");
   * out = icm -> model . Device_Model ();
   return  GMG_OK;
  }
