//  icm.cc -- host side of the MI355X-native ICM_t (see icm.hh).
//
//  Model I/O (reference: src/ICM/icm.cc:614-803,846-861,961-998) and the
//  null-model builder (src/ICM/icm.cc:65-350) run on the host, as in the
//  reference.  Every method that returns a score packs its string, hands it to
//  the extern "C" HIP layer (include/gmg.h) and copies the result back; a
//  failure of that layer is fatal with a message on stderr, matching the
//  reference's exit(EXIT_FAILURE) convention (src/ICM/icm.cc:635-652).
//  The one-call-one-launch methods exist for interface compatibility; bulk
//  work belongs on the batch entry points of include/gmg.h.

#include "icm.hh"
#include "../../include/gmg.h"

#include <assert.h>
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

using namespace std;

extern int  Verbose;   // src/Common/delcher.cc:20 in the reference; weak default in gmg_icm_c.cc

// ---------------------------------------------------------------------------
// small helpers around the C ABI
// ---------------------------------------------------------------------------

namespace {

void  Device_Fatal  (const char * who)
  {
   fprintf (stderr, "ERROR:  %s: %s\n", who, gmg_last_error ());
   exit (EXIT_FAILURE);
  }

void  Ensure_Device  (void)
  {
   static bool  ready = false;
   if  (ready)
       return;
   const char  * env = getenv ("GMG_DEVICE");
   if  (gmg_init (env ? atoi (env) : 0) != GMG_OK)
       Device_Fatal ("gmg_init");
   ready = true;
  }

//  The calling thread's staging (page-locked host buffer + device buffers of gmg_single), made at the thread's first call and
//  freed when the thread ends (a thread_local object's destructor)
struct  Thread_Staging_t
  {
   gmg_single  * st;
   Thread_Staging_t  ()  : st (NULL)  {}
   ~ Thread_Staging_t  ()  { if  (st != NULL)  gmg_single_free (st); }
  };
static gmg_single  * Thread_Staging  (void)
  {
   static thread_local Thread_Staging_t  mine;
   if  (mine . st == NULL && gmg_single_create (& mine . st) != GMG_OK)
       Device_Fatal ("gmg_single_create");
   return  mine . st;
  }

//  One string as a one-read batch in HBM, plus a device output buffer: the calling thread's gmg_single (persistent
//  page-locked staging and device buffers: a call is one copy in, one launch, one copy out; nothing is allocated)
struct  One_Read_t
  {
   const gmg_reads  * reads;
   const gmg_segments  * segs;
   double  * d_out;
   gmg_single  * stage;

   One_Read_t  (const char * s, int n, gmg_orient orient, size_t out_doubles)
     : reads (NULL), segs (NULL), d_out (NULL), stage (NULL)
     {
      Ensure_Device ();
      stage = Thread_Staging ();
      (void) out_doubles;                                //  (the staging has room for n + 16)
      if  (gmg_single_stage (stage, s, (uint64_t) n, (int) orient, & reads, & segs, & d_out) != GMG_OK)
          Device_Fatal ("ICM_t device staging");
     }
   void  Fetch  (double * dst, size_t n)
     {
      if  (n > 0 && gmg_single_fetch (stage, dst, n) != GMG_OK)
          Device_Fatal ("ICM_t result copy");
     }
  };

int  Power_Of_4  (int e)
  {
   int  r = 1;
   while  (e -- > 0)
     r *= 4;
   return  r;
  }

}  // namespace


// ---------------------------------------------------------------------------
// construction / destruction
// ---------------------------------------------------------------------------

void  ICM_t :: Alloc_Tables  (void)
  {
   score = (ICM_Score_Node_t * *) calloc (periodicity > 0 ? periodicity : 1, sizeof (ICM_Score_Node_t *));
   if  (score == NULL)
       {
        fprintf (stderr, "ERROR:  calloc failed for ICM table\n");
        exit (EXIT_FAILURE);
       }
   for  (int i = 0;  i < periodicity;  i ++)
     {
      score [i] = (ICM_Score_Node_t *) calloc (num_nodes > 0 ? num_nodes : 1, sizeof (ICM_Score_Node_t));
      if  (score [i] == NULL)
          {
           fprintf (stderr, "ERROR:  calloc failed for ICM table\n");
           exit (EXIT_FAILURE);
          }
     }
  }

void  ICM_t :: Free_Tables  (void)
  {
   if  (score != NULL)
       {
        for  (int i = 0;  i < periodicity;  i ++)
          free (score [i]);
        free (score);
        score = NULL;
       }
  }

//  src/ICM/icm.cc:24-44
ICM_t :: ICM_t  (int w, int d, int p)
  {
   model_len = w;
   model_depth = d;
   periodicity = p;
   num_nodes = (Power_Of_4 (model_depth + 1) - 1) / (ALPHABET_SIZE - 1);
   dev_mirror = NULL;
   Alloc_Tables ();
   empty = true;
  }

//  src/ICM/icm.cc:48-61
ICM_t :: ~ ICM_t  ()
  {
   Invalidate_Device_Mirror ();
   Free_Tables ();
  }

//  src/ICM/icm.cc:1000-1007: shallow copy, tables shared (never destroy both)
void  ICM_t :: Copy  (ICM_t & icm)
  {
   Invalidate_Device_Mirror ();
   empty = icm . empty;
   model_len = icm . model_len;
   model_depth = icm . model_depth;
   periodicity = icm . periodicity;
   num_nodes = icm . num_nodes;
   score = icm . score;
  }


// ---------------------------------------------------------------------------
// device mirror
// ---------------------------------------------------------------------------

void  ICM_t :: Export_Tables  (vector <short> & mip, vector <float> & prob4)  const
  {
   size_t  pn = size_t (periodicity) * num_nodes;
   mip . resize (pn);
   prob4 . resize (4 * pn);
   for  (int p = 0;  p < periodicity;  p ++)
     for  (int n = 0;  n < num_nodes;  n ++)
       {
        size_t  slot = size_t (p) * num_nodes + n;
        mip [slot] = score [p] [n] . mut_info_pos;
        memcpy (& prob4 [4 * slot], score [p] [n] . prob, 4 * sizeof (float));
       }
  }

const gmg_model *  ICM_t :: Device_Model  (void)  const
  {
   if  (dev_mirror == NULL)
       {
        vector <short>  mip;
        vector <float>  prob4;
        Ensure_Device ();
        Export_Tables (mip, prob4);
        if  (gmg_model_upload (mip . data (), prob4 . data (), model_len, model_depth,
                               periodicity, num_nodes, & dev_mirror) != GMG_OK)
            Device_Fatal ("gmg_model_upload");
       }
   return  dev_mirror;
  }

void  ICM_t :: Invalidate_Device_Mirror  (void)  const
  {
   if  (dev_mirror != NULL)
       {
        gmg_model_free (dev_mirror);
        dev_mirror = NULL;
       }
  }


// ---------------------------------------------------------------------------
// model I/O
// ---------------------------------------------------------------------------

//  Binary format (src/ICM/icm.cc:614-726): 150-byte text header; six int32
//  {version 200, 150, model_len, depth, periodicity, num_nodes}; records
//  {int32 id, float prob[4], int16 mut_info_pos}; id 0 starts the next
//  sub-model; ids that never appear are cut nodes (mut_info_pos = -2);
//  a negative id ends the stream.
bool  ICM_t :: Try_Input  (FILE * fp, string & err)
  {
   char  header [ID_STRING_LEN];
   int  param [NUM_FIXED_LENGTH_PARAMS];
   char  msg [200];

   Invalidate_Device_Mirror ();
   Free_Tables ();

   if  (fread (header, 1, ID_STRING_LEN, fp) != size_t (ID_STRING_LEN))
       { err = "ERROR reading ICM header";  return  false; }
   if  (fread (param, sizeof (int), NUM_FIXED_LENGTH_PARAMS, fp) != NUM_FIXED_LENGTH_PARAMS)
       { err = "ERROR reading parameters";  return  false; }
   if  (param [0] != ICM_VERSION_ID)
       {
        snprintf (msg, sizeof msg, "Bad ICM version = %d  should be %d", param [0], ICM_VERSION_ID);
        err = msg;  return  false;
       }
   if  (param [1] != ID_STRING_LEN)
       {
        snprintf (msg, sizeof msg, "Bad ID_STRING_LEN = %d  should be %d", param [1], ID_STRING_LEN);
        err = msg;  return  false;
       }
   if  (param [2] <= 0 || param [3] < 0 || param [4] <= 0 || param [5] <= 0)
       { err = "ERROR:  bad ICM parameters";  periodicity = 0;  return  false; }

   model_len = param [2];
   model_depth = param [3];
   periodicity = param [4];
   num_nodes = param [5];
   Alloc_Tables ();

   vector < vector <bool> >  present (periodicity, vector <bool> (num_nodes, false));
   int  period = -1, node_id;
   while  (fread (& node_id, sizeof (int), 1, fp) == 1 && node_id >= 0)
     {
      if  (node_id == 0)
          period ++;
      if  (period < 0 || period >= periodicity || node_id >= num_nodes)
          {
           snprintf (msg, sizeof msg, "ERROR reading icm node = %d  period = %d", node_id, period);
           err = msg;  return  false;
          }
      ICM_Score_Node_t  & nd = score [period] [node_id];
      if  (fread (nd . prob, sizeof (float), ALPHABET_SIZE, fp) != size_t (ALPHABET_SIZE))
          {
           snprintf (msg, sizeof msg, "ERROR reading icm node = %d  period = %d", node_id, period);
           err = msg;  return  false;
          }
      if  (fread (& nd . mut_info_pos, sizeof (short int), 1, fp) != 1)
          {
           snprintf (msg, sizeof msg, "ERROR reading mut_info_pos for node = %d  period = %d", node_id, period);
           err = msg;  return  false;
          }
      present [period] [node_id] = true;
     }
   if  (period != periodicity - 1)
       {
        snprintf (msg, sizeof msg, "ERROR:  Too few nodes for periodicity = %d", periodicity);
        err = msg;  return  false;
       }
   //  every id the stream skipped is a cut node.  (The reference marks gaps as it
   //  reads, icm.cc:699-721; with ids in increasing order, which its writer always
   //  produces, that is exactly "absent => -2" except for the never-absent root.)
   for  (int p = 0;  p < periodicity;  p ++)
     for  (int n = 1;  n < num_nodes;  n ++)
       if  (! present [p] [n])
           score [p] [n] . mut_info_pos = -2;

   empty = false;
   return  true;
  }

void  ICM_t :: Input  (FILE * fp)
  {
   string  err;
   if  (! Try_Input (fp, err))
       {
        fprintf (stderr, "%s\n", err . c_str ());
        exit (EXIT_FAILURE);
       }
  }

//  src/ICM/icm.cc:846-861 (open failure: src/Common/delcher.cc:90-111)
void  ICM_t :: Read  (char * path)
  {
   FILE  * fp = fopen (path, "r");
   if  (fp == NULL)
       {
        fprintf (stderr, "ERROR:  Could not open file  %s\n  errno = %d\n", path, errno);
        exit (EXIT_FAILURE);
       }
   Input (fp);
   fclose (fp);
  }

//  src/ICM/icm.cc:961-998
void  ICM_t :: Write_Header  (FILE * fp, bool binary_form)
  {
   if  (! binary_form)
       {
        fprintf (fp, "ver = %.2f  len = %d  depth = %d  periodicity = %d  nodes = %d\n",
                 ICM_VERSION_ID / 100.0, model_len, model_depth, periodicity, num_nodes);
        return;
       }
   char  line [ID_STRING_LEN];
   int  param [NUM_FIXED_LENGTH_PARAMS] = {ICM_VERSION_ID, ID_STRING_LEN, model_len,
                                           model_depth, periodicity, num_nodes};
   memset (line, 0, sizeof line);
   snprintf (line, sizeof line, ">ver = %.2f  len = %d  depth = %d  periodicity = %d  nodes = %d\n",
             ICM_VERSION_ID / 100.0, model_len, model_depth, periodicity, num_nodes);
   fwrite (line, 1, ID_STRING_LEN, fp);
   fwrite (param, sizeof (int), NUM_FIXED_LENGTH_PARAMS, fp);
  }

//  src/ICM/icm.cc:757-803
void  ICM_t :: Output_Node
    (FILE * fp, ICM_Score_Node_t * node, int id, int frame, bool binary_form)
  {
   if  (Verbose > 1)
       fprintf (stderr, "output node %d  frame %d\n", id, frame);

   if  (binary_form)
       {
        fwrite (& id, sizeof (int), 1, fp);
        fwrite (node -> prob, sizeof (float), ALPHABET_SIZE, fp);
        fwrite (& node -> mut_info_pos, sizeof (short int), 1, fp);
        return;
       }

   char  label [2 * 100];
   assert (model_len <= 100);
   memset (label, '-', model_len);
   label [model_len - 1] = '?';
   label [model_len] = '\0';
   Set_Label_String (label, id, frame);
   if  (Verbose > 1)
       fprintf (stderr, "Label set to %s\n", label);
   fprintf (fp, "%6d  %s", id, label);
   fprintf (fp, " %7.4f", node -> mut_info);
   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     fprintf (fp, " %6.3f", exp (node -> prob [i]));
   fputc ('\n', fp);
  }

//  src/ICM/icm.cc:729-753: the root of each sub-model always, other nodes when present
void  ICM_t :: Output  (FILE * fp, bool binary_form)
  {
   Write_Header (fp, binary_form);
   for  (int frame = 0;  frame < periodicity;  frame ++)
     for  (int i = 0;  i < num_nodes;  i ++)
       if  (i == 0 || score [frame] [i] . mut_info_pos >= -1)
           Output_Node (fp, score [frame] + i, i, frame, binary_form);
   if  (binary_form)
       {
        int  end_marker = -1;
        fwrite (& end_marker, sizeof (int), 1, fp);
       }
  }

//  src/ICM/icm.cc:455-482
void  ICM_t :: Display  (FILE * fp)
  {
   fprintf (fp, "model_len = %d  periodicity = %d  depth = %d  num_nodes = %d\n",
            model_len, periodicity, model_depth, num_nodes);
   for  (int period = 0;  period < periodicity;  period ++)
     {
      fprintf (fp, "period = %d\n", period);
      for  (int i = 0;  i < num_nodes;  i ++)
        {
         fprintf (fp, "%3d:  %2d ", i, score [period] [i] . mut_info_pos);
         for  (int j = 0;  j < ALPHABET_SIZE;  j ++)
           fprintf (fp, " %7.4f", exp (score [period] [i] . prob [j]));
         fputc ('\n', fp);
        }
     }
  }

//  src/ICM/icm.cc:907-957: mark the context positions fixed by the ancestors of
//  node  id , then spread the label so that  '|'  separates codon-sized groups.
void  ICM_t :: Set_Label_String  (char * label, int id, int frame)
  {
   int  mip = score [frame] [id] . mut_info_pos;
   if  (mip >= 0)
       label [mip] = MAX_MI_CHAR;
   for  (int child = id;  child > 0;  )
     {
      int  up = PARENT (child);
      label [score [frame] [up] . mut_info_pos] = ALPHA_STRING [child - ALPHABET_SIZE * up - 1];
      child = up;
     }

   int  last_sep = 0, sep_ct = 0;
   if  (periodicity != 1)
       {
        last_sep = (frame == 0) ? model_len - periodicity : model_len - frame;
        if  (last_sep < 0)
            last_sep = 0;
        sep_ct = (last_sep + periodicity - 1) / periodicity;
       }
   for  (int i = model_len;  i > 0;  i --)
     {
      label [i + sep_ct] = label [i];
      if  (i == last_sep)
          {
           sep_ct --;
           label [i + sep_ct] = SEPARATOR_CHAR;
           last_sep -= periodicity;
          }
     }
  }


// ---------------------------------------------------------------------------
// null model
// ---------------------------------------------------------------------------

//  Shared tail of Build_Indep_WO_Stops / Build_Reverse_Codon_WO_Stops
//  (src/ICM/icm.cc:116-216 and 250-350): knock out the stop codons (spelled
//  backwards: ORFs are scored 3' to 5'), renormalise, marginalise the 64 codon
//  probabilities into the three 21-node trees, take logs.  The sums are kept in
//  the float  prob  fields, as the reference does.
void  ICM_t :: Build_From_Codon_Probs
    (double codon_prob [64], const vector <const char *> & stop_codon, const char * who)
  {
   if  (model_len != 3 || model_depth != 2 || periodicity != 3 || num_nodes != 21)
       {
        fprintf (stderr, "ERROR:  Incompatible ICM_Training_t for %s\n", who);
        fprintf (stderr, "model_len = %d  model_depth = %d  periodicity = %d\n"
                         "alphabet_size = %d  num_nodes = %d\n",
                 model_len, model_depth, periodicity, ALPHABET_SIZE, num_nodes);
        fprintf (stderr, "Should be  %d ,  %d ,  %d ,  %d ,  %d  respectively\n", 3, 2, 3, 4, 21);
        exit (EXIT_FAILURE);
       }
   Invalidate_Device_Mirror ();

   for  (size_t i = 0;  i < stop_codon . size ();  i ++)
     {
      const char  * c = stop_codon [i];
      codon_prob [Subscript (c [0]) + 4 * Subscript (c [1]) + 16 * Subscript (c [2])] = 1e-20;
     }
   double  total = 0.0;
   for  (int j = 0;  j < 64;  j ++)
     total += codon_prob [j];
   for  (int j = 0;  j < 64;  j ++)
     codon_prob [j] /= total;

   for  (int f = 0;  f < 3;  f ++)
     for  (int n = 0;  n < 21;  n ++)
       {
        score [f] [n] . mut_info = 0.0;
        for  (int k = 0;  k < 4;  k ++)
          score [f] [n] . prob [k] = 0.0;
       }

   //  digit of codon index j that sub-model f predicts / conditions on
   static const int  place [3] = {1, 4, 16};
   for  (int f = 0;  f < 3;  f ++)
     {
      const int  d1 = place [(3 - f) % 3], d2 = place [(4 - f) % 3];
      ICM_Score_Node_t  * root = score [f];
      root -> mut_info_pos = (f == 1) ? -1 : 1;
      for  (int j = 0;  j < 64;  j ++)
        root -> prob [(j / d1) % 4] += codon_prob [j];

      ICM_Score_Node_t  * lvl1 = score [f] + 1;
      for  (int b = 0;  b < 4;  b ++)
        lvl1 [b] . mut_info_pos = (f == 2) ? -1 : 0;
      if  (f != 1)
          for  (int j = 0;  j < 64;  j ++)
            lvl1 [(j / d2) % 4] . prob [(j / d1) % 4] += codon_prob [j];
     }
   ICM_Score_Node_t  * lvl2 = score [0] + 5;      // only sub-model 0 reaches level 2
   for  (int k = 0;  k < 16;  k ++)
     lvl2 [k] . mut_info_pos = -1;
   for  (int j = 0;  j < 64;  j ++)
     lvl2 [4 * ((j / 4) % 4) + (j / 16) % 4] . prob [j % 4] += codon_prob [j];

   for  (int f = 0;  f < 3;  f ++)
     for  (int n = 0;  n < 21;  n ++)
       {
        float  * p = score [f] [n] . prob;
        double  sum = 0.0;
        for  (int k = 0;  k < 4;  k ++)
          sum += p [k];
        for  (int k = 0;  k < 4;  k ++)
          p [k] = (sum == 0.0 ? 0.0 : log (p [k] / sum));
       }
   empty = false;
  }

//  src/ICM/icm.cc:65-114 then the shared tail
void  ICM_t :: Build_Indep_WO_Stops
    (double gc_frac, const vector <const char *> & stop_codon)
  {
   double  codon_prob [64], base_prob [4];
   base_prob [1] = base_prob [2] = gc_frac / 2.0;
   base_prob [0] = base_prob [3] = 0.5 - base_prob [1];
   for  (int j = 0;  j < 64;  j ++)
     codon_prob [j] = base_prob [(j >> 4) & 3] * base_prob [(j >> 2) & 3] * base_prob [j & 3];
   Build_From_Codon_Probs (codon_prob, stop_codon, "Build_Indep_WO_Stops");
  }

//  src/ICM/icm.cc:220-350 (no callers in the reference)
void  ICM_t :: Build_Reverse_Codon_WO_Stops
    (double codon_prob [64], const vector <const char *> & stop_codon)
  {
   Build_From_Codon_Probs (codon_prob, stop_codon, "Build_Reverse_Codon_WO_Stops");
  }


// ---------------------------------------------------------------------------
// scoring: every method below is one or two launches of the HIP layer
// ---------------------------------------------------------------------------

//  src/ICM/icm.cc:354-405
void  ICM_t :: Cumulative_Score
    (const string & s, vector <double> & out, int frame)  const
  {
   if  (periodicity == 1)
       frame = 0;
   assert (0 <= frame && frame < periodicity);
   int  n = s . length ();
   out . resize (n);
   if  (n == 0)
       return;
   One_Read_t  job (s . c_str (), n, GMG_FORWARD, n);
   if  (gmg_segment_cumscore (Device_Model (), job . reads, job . segs, frame,
                              (double *) job . d_out, NULL) != GMG_OK)
       Device_Fatal ("gmg_segment_cumscore");
   job . Fetch (& out [0], n);
  }

//  src/ICM/icm.cc:409-452: cum_score [0] = 0, entry i+1 = sum through base i.
//  The reference always scores the first model_len-1 bases, so  len  must be at
//  least that (it would read past the string otherwise).
void  ICM_t :: Cumulative_Score_String
    (char * string, int len, int frame, double * cum_score)
  {
   if  (periodicity == 1)
       frame = 0;
   assert (0 <= frame && frame < periodicity);
   if  (Verbose > 0)
       printf ("Cumulative_Score_String  len = %d  frame = %d\n", len, frame);
   cum_score [0] = 0.0;
   int  n = (len > model_len - 1) ? len : model_len - 1;
   if  (n == 0)
       return;
   One_Read_t  job (string, n, GMG_FORWARD, n);
   if  (gmg_segment_cumscore (Device_Model (), job . reads, job . segs, frame,
                              (double *) job . d_out, NULL) != GMG_OK)
       Device_Fatal ("gmg_segment_cumscore");
   job . Fetch (cum_score + 1, n);
  }

//  src/ICM/icm.cc:485-509
void  ICM_t :: Frame_Score
    (const string & s, vector <double> & out, int frame)  const
  {
   assert (0 <= frame && frame < periodicity);
   int  n = s . length ();
   out . resize (n);
   if  (n == 0)
       return;
   One_Read_t  job (s . c_str (), n, GMG_FORWARD, n);
   if  (gmg_segment_frame_score (Device_Model (), job . reads, job . segs, frame,
                                 (double *) job . d_out, NULL) != GMG_OK)
       Device_Fatal ("gmg_segment_frame_score");
   job . Fetch (& out [0], n);
  }

//  src/ICM/icm.cc:864-903
double  ICM_t :: Score_String
    (const char * string, int len, int frame)  const
  {
   if  (periodicity == 1)
       frame = 0;
   assert (0 <= frame && frame < periodicity);
   if  (Verbose > 0)
       printf ("Score_String  len = %d  frame = %d\n", len, frame);
   if  (len <= 0)
       return  0.0;
   double  result;
   One_Read_t  job (string, len, GMG_FORWARD, 1);
   if  (gmg_score_string (Device_Model (), job . reads, job . segs, frame,
                          (double *) job . d_out, NULL) != GMG_OK)
       Device_Fatal ("gmg_score_string");
   job . Fetch (& result, 1);
   return  result;
  }

namespace {

//  one explicit window through gmg_window_distrib, on the calling thread's staging (nothing is allocated per call)
void  One_Window
    (const gmg_model * m, const char * string, int model_len, int frame, float * dist, double * prob)
  {
   vector <uint8_t>  codes (model_len);
   for  (int k = 0;  k < model_len;  k ++)
     codes [k] = (uint8_t) gmg_base_code ((unsigned char) string [k]);
   if  (gmg_single_window (Thread_Staging (), m, codes . data (), model_len, frame, dist, prob) != GMG_OK)
       Device_Fatal ("gmg_window_distrib");
  }

}  // namespace

//  src/ICM/icm.cc:512-553
void  ICM_t :: Full_Window_Distrib
    (char * string, int frame, float * dist)
  {
   Ensure_Device ();
   One_Window (Device_Model (), string, model_len, frame, dist, NULL);
  }

//  src/ICM/icm.cc:557-610
double  ICM_t :: Full_Window_Prob
    (const char * string, int frame)  const
  {
   double  prob;
   Ensure_Device ();
   One_Window (Device_Model (), string, model_len, frame, NULL, & prob);
   return  prob;
  }

//  src/ICM/icm.cc:807-842.  The partial-window rule is applied whatever
//  predict_pos  is, as in the reference.
double  ICM_t :: Partial_Window_Prob
    (int predict_pos, const char * string, int frame)  const
  {
   double  prob;
   One_Read_t  job (string, predict_pos + 1, GMG_FORWARD, 1);
   if  (gmg_segment_partial_prob (Device_Model (), job . reads, job . segs, frame,
                                  (double *) job . d_out, NULL) != GMG_OK)
       Device_Fatal ("gmg_segment_partial_prob");
   job . Fetch (& prob, 1);
   return  prob;
  }


//  src/ICM/icm.cc:2008-2027
int  Subscript  (char ch)
  {
   //  tolower (Filter (ch)) always lands in "acgt" (src/Common/gene.cc:1139-1175), so the
   //  reference's "Bad character" exit is unreachable; gmg_base_code is that composition.
   return  gmg_base_code ((unsigned char) ch);
  }
